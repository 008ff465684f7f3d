"""``data.cvs`` of the reference."""
