"""Stereo camera model, triangulation and PnP-RANSAC facades."""
