"""Synthetic data and small helpers."""
