"""Data side of the hot path: the on-device input pipeline."""
