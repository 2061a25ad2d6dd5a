"""Process-wide opt-in switches of the MI355X path (extensions beyond the reference's behaviour)."""

# Run eval-mode quantized Linears with fixed ranges as exact integer GEMMs on the i8 matrix cores with
# bias / activation / output quantizer fused into the epilogue (tq_linear_i8_fwd); quantizers with a
# fixed per-tensor range then also emit their int8 grid indices in the same launch so that the GEMM
# can consume them directly.  Off by default: the default path reproduces the reference's fp32
# simulation; the integer path evaluates the same numbers exactly and therefore differs from the
# simulation by the simulation's own fp32 accumulation error (~1e-6 relative), which can move an
# output that sits on a rounding boundary by one grid step.
INT8_LINEAR = False

# Integer Linears with a GELU: evaluate activation + output quantizer through a staircase table built on the device from
# the quantizer's range (csrc/tq_stair.hip, one extra launch per range state) instead of the erf fit + exact quotient in
# the epilogue.  The table is the correctly rounded GELU followed by the reference quantizer, exact by construction; the
# arithmetic epilogue stays in place for grids the table cannot hold (decided on the device).
INT8_ACT_STAIR = True

# Estimator state (current_xmin / current_xmax) and quantizer parameters (_delta / _zero_float / _signed)
# are rebound to FRESH tensors on every calibrating forward, like the reference does.  With this switch
# the fused calibration step updates the existing buffers IN PLACE once they exist (same values, same
# shapes), which makes a calibrating forward replayable as a hipGraph: capture one forward after the
# first (eager) batch, then copy each new batch into the static input and replay -- no Python, no
# allocation, no host synchronisation between the 161 + 102 quantizers of a BERT-base.  Off by
# default because code that keeps references to earlier state tensors would see them change.
INPLACE_CALIBRATION_STATE = False

# AdaRound: after three eager iterations the loop body (sample gather, soft-quantized weight, layer forward, gradient
# GEMM, fused backward + regulariser + Adam) is recorded once as a hipGraph and replayed for the remaining iterations,
# with the per-iteration sample indices and schedule scalars read from device tables.  Same kernels in the same order
# on the same data: the learned rounding is bit-identical to the eager loop (tests/test_adaround_inits.py).  Applies to
# the fused step of plain Linear layers (closed-form gradient, no folded activation function) on one GPU with the
# relaxation loss; anything else runs the eager loop.
GRAPH_ADAROUND = True

# Bumped by anything that rewrites parameters or range buffers behind autograd's back -- a hipGraph replay of a training
# step updates weights and learnable ranges in place WITHOUT touching tensor._version -- so that every derived cache
# (int8 weight indices, NoNorm parameters, stacked QKV operands, provenance records) sees a new key:
# QuantizerBase.range_state_key() includes it.
CACHE_EPOCH = 0


def invalidate_derived_caches():
    global CACHE_EPOCH
    CACHE_EPOCH += 1
