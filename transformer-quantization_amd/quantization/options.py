"""Process-wide opt-in switches of the MI355X path (extensions beyond the reference's behaviour)."""

# Run eval-mode quantized Linears with fixed ranges as exact integer GEMMs on the i8 matrix cores with
# bias / activation / output quantizer fused into the epilogue (tq_linear_i8_fwd); quantizers with a
# fixed per-tensor range then also emit their int8 grid indices in the same launch so that the GEMM
# can consume them directly.  Off by default: the default path reproduces the reference's fp32
# simulation; the integer path evaluates the same numbers exactly and therefore differs from the
# simulation by the simulation's own fp32 accumulation error (~1e-6 relative), which can move an
# output that sits on a rounding boundary by one grid step.
INT8_LINEAR = False
