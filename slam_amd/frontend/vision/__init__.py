"""Descriptor matching facade."""
