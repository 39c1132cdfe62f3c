"""Top-level `model` package of the reference layout, re-exporting the MI355X-native mirrors."""
