"""Process-wide switches of the MI355X path (extensions beyond the reference's behaviour)."""
import torch

# Run quantized Linears with fixed ranges as exact integer GEMMs on the i8 matrix cores with bias / activation / output
# quantizer fused into the epilogue (tq_linear_i8_fwd); quantizers with a fixed per-tensor range then also emit their int8
# grid indices in the same launch so that the GEMM can consume them directly, and the harness models run their fused
# fixed-range tails / attention cores.
#   'auto' (default): whenever autograd is off (torch.no_grad()) -- the fixed-range EVALUATION forward (and, for large GEMMs,
#           calibrating forwards: INT8_CALIBRATION below).
#           Training and anything with forward hooks on the modules involved keep the layered route (one launch per
#           quantizer around torch's fp32 GEMM, the reference's module chain); so do the fused tails / attention core / small
#           GEMMs of a calibrating forward (their quantizers are still estimating).
#   True:   also under autograd (QAT forward on the matrix cores, straight-through backward).
#   False:  never -- the layered route everywhere.
# Why 'auto' is the default (round 5, profiles/r05/int_vs_reference.json, tests/test_bert_e2e.py / test_mobilebert_e2e.py
# `..._default_route_...`): against the REFERENCE's own outputs the integer route is as close as the layered GPU route --
# the layered route's hipBLASLt GEMMs differ from the reference's CPU GEMMs by fp32 round-off just as the exact integer
# contraction does, and through 12-24 quantized layers either difference is amplified the same way (BERT-base W8A8,
# reference ranges installed: 24.67 % vs 24.56 % of the last layer's 786 432 outputs on the reference's grid point, mean
# deviation 1.283 vs 1.291 steps; first layer 99.80 % vs 99.20 %) -- while it is deterministic, exact arithmetic and 4x
# faster (3.26 -> 0.80 ms).
INT8_LINEAR = 'auto'

# Calibrating forwards (ranges still being estimated, autograd off) on the integer route as well: a quantized Linear whose
# INPUT was produced by a quantizer that has just set its range for this batch (per-tensor asymmetric <= 8 bit) and whose
# weight grid is known runs the exact integer GEMM (bias + activation function in the epilogue, fp32 result to the output
# quantizer's estimator) instead of the fp32 simulation -- the GEMMs are 2 of the 4.2 ms of a BERT-base calibrating forward
# at [8,128] and 25 of 33 ms at [128,128].  The statistics each estimator sees are those of the EXACT products the fp32
# GEMMs approximate (ranges differ from the layered GPU route's by its round-off, as the layered GPU route's differ from
# the reference's CPU GEMMs); everything else -- estimators, state machine, exchanges of a sharded calibration -- is the
# layered code.  Follows INT8_LINEAR: off where that is off (and under autograd / training mode / forward hooks).
INT8_CALIBRATION = True
# ... for GEMMs of at least this many multiply-accumulates (M x K x N; 2^33 = 16 384 tokens through a 768 x 768 Linear, 4 096
# through BERT's feed-forward Linears).  Below, the ~45 us of extra host work + one more launch per layer (the input's
# indices) cost an EAGER calibrating forward more than the fp32 GEMM they replace (profiles/r06/calib_int8_ab.txt: with
# every Linear on the integer path [8,128] went 7.8 -> 12.2 ms eager while its hipGraph went 4.2 -> 3.2 ms; [128,128]:
# 33.1 -> 19.1 ms either way).  The rule depends on the shapes only, so a recorded forward takes the route the eager one took.
INT8_CALIBRATION_MIN_MACS = 1 << 33

# Integer Linears with a GELU: evaluate activation + output quantizer through a staircase table built on the device from
# the quantizer's range (csrc/tq_stair.hip, one extra launch per range state) instead of the erf fit + exact quotient in
# the epilogue.  The table is the correctly rounded GELU followed by the reference quantizer, exact by construction; the
# arithmetic epilogue stays in place for grids the table cannot hold (decided on the device).
# Reproducibility note: the table evaluates the CORRECTLY ROUNDED GELU, the arithmetic epilogue a degree-7 erfc fit; they
# agree on all but <= 2e-5 of the outputs, where the index differs by ONE grid step.  Which of the two a launch uses depends
# on the tile kernel (call shape: M, N), on the room for the table and on the builder's verdict -- so the same layer with
# the same ranges can differ in those few indices between batch sizes.  Every launch is deterministic; set
# INT8_ACT_STAIR = False for one definition (the fit) at every shape.
INT8_ACT_STAIR = True

# README recipe (MSE / golden-section weight ranges, reference README.md:149-157): run the searches of ALL weight tensors in
# lock step before the first calibrating forward (autoquant_utils.precalibrate_weights -> range_estimators.
# golden_section_lockstep): every search is scipy's bounded Brent restated as a resumable generator (pinned against scipy,
# tests/test_lockstep.py), each round evaluates the pending candidate of every live search with prepared launches queued
# back to back and ONE device->host copy.  Bit-identical ranges; BERT-base (102 tensors, 2 096 evaluations in 26 rounds):
# 148-156 ms layer by layer (71-75 us per evaluation: 11 us of kernels, 17 us of synchronisation, ~25 us of Python) ->
# 64.5 ms (2.3x; the kernels alone are ~38 ms).  (A first version with one scipy instance per search in its own THREAD was
# slower than layer by layer -- 250-290 ms: 102 thread wake-ups per round -- and is gone.)
LOCKSTEP_WEIGHT_SEARCH = True

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


def int8_active():
    """Is the integer / fused fixed-range route on for the call being made right now?  (see INT8_LINEAR)"""
    m = INT8_LINEAR
    if not m:
        return False
    # torch.inference_mode(): tensors created there carry no version counter, so an in-place change of an activation between
    # the quantizer that produced it and the integer Linear that consumes its indices could not be noticed (provenance.py):
    # the layered route runs there, whatever the switch says.  torch.no_grad() keeps the counters and is what 'auto' is for.
    if torch.is_inference_mode_enabled():
        return False
    if m is True:
        return True
    return not torch.is_grad_enabled()


def fuse_on(flag, module=None, probe=None):
    """Tri-state `fuse` switches of the harness models: True / False force it, None (default) follows INT8_LINEAR for
    modules in eval mode whose ranges are fixed.  probe: a QuantizedActivation / quantized layer of the block whose
    manager's state stands for the block's (a calibrating forward then skips the fused helpers' eligibility checks
    altogether instead of failing them one by one: ~1 ms of host time per BERT-base forward)."""
    if flag is None:
        if not int8_active() or (module is not None and module.training):
            return False
        if probe is not None:
            mgr = probe._modules.get('activation_quantizer')
            state = getattr(mgr, 'state', None)
            return state is None or state.name == 'fix_ranges'
        return True
    return bool(flag)


def invalidate_derived_caches():
    global CACHE_EPOCH
    CACHE_EPOCH += 1
