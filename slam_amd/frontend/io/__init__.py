"""Frame and feature input."""
