"""Front-end: matching, stereo geometry and PnP behind the reference class names."""
