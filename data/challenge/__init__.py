"""``data.challenge`` of the reference."""
