"""Top-level `utils` package of the reference layout, re-exporting the MI355X-native mirrors."""
