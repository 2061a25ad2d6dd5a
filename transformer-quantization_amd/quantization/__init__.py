"""MI355X-native drop-in for the reference's ``quantization`` package (fake-quant hot path)."""
