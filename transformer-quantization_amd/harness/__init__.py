"""Synthetic-data consumers of the drop-in quantization package (no checkpoints / datasets offline)."""
