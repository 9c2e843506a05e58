"""Loop-closure candidate verification."""
