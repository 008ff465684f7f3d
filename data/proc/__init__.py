"""``data.proc`` of the reference."""
