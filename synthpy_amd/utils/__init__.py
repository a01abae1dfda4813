"""File formats either side of the path (src/utils of the reference)."""
