"""Hand-over to the back-end: tracking records and loop verification."""
