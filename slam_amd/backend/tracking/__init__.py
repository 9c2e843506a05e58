"""Tracking records, feature cache and the tracking-database adaptor."""
