"""Which quantizer produced this tensor -- and, if the producing kernel emitted them, its int8 grid indices.

The integer fast paths (MFMA Linear, integer attention core, fused tails; `options.INT8_LINEAR`) consume activations as
int8 grid indices.  A consumer may only do so if it KNOWS the tensor lies on a fixed per-tensor grid: that knowledge is
recorded here by the producer (`tag`) and looked up by the consumer (`of`), in one explicit table instead of ad-hoc
tensor attributes:

    record = (quantizer, idx)      quantizer: the fixed-range quantizer whose grid the values lie on
                                   idx: int8(index - 128) of the same shape, or None (the consumer re-derives them
                                        exactly with one tq_fake_quant_fwd index-only launch)

Records are keyed by tensor OBJECT identity and die with the tensor (weak references).  Anything that makes a new
tensor object -- a view, `.contiguous()`, dropout, an arithmetic op -- has no record, and the consumer takes the
layered path (always correct, only slower).  That is the intended safety property, not an accident: values are trusted
to be on-grid only for the very object the quantizer kernel returned -- and only while that object is UNCHANGED: a
record also carries the tensor's version counter (any in-place op: `add_`, in-place dropout, `copy_`) and the range
state of the producing quantizer (a range re-set between producer and consumer), and is ignored once either moved.
"""
import weakref

_records = {}


def version_of(tensor):
    """The tensor's in-place modification counter, or None where torch does not keep one (inference tensors)."""
    try:
        return tensor._version
    except RuntimeError:
        return None


def tag(tensor, quantizer, idx=None):
    """Record that `tensor` was produced by `quantizer` (fixed range); returns the tensor.  A tensor without a version
    counter (created under torch.inference_mode()) gets NO record: an in-place change could not be detected later, so its
    consumers take the layered path."""
    key = id(tensor)
    if version_of(tensor) is None:
        _records.pop(key, None)
        return tensor

    def _drop(_ref, key=key):
        _records.pop(key, None)
    _records[key] = (weakref.ref(tensor, _drop), quantizer, idx, tensor._version, _range_state(quantizer))
    return tensor


def _range_state(quantizer):
    key = getattr(quantizer, 'range_state_key', None)
    return key() if key is not None else None


def of(tensor):
    """(quantizer, idx | None) recorded for this tensor object, or None (no record, or a stale one)."""
    rec = _records.get(id(tensor))
    if rec is None or rec[0]() is not tensor:
        return None
    if version_of(tensor) != rec[3] or _range_state(rec[1]) != rec[4]:
        return None                          # modified in place / producer's grid changed since the record was made
    return rec[1], rec[2]


def quantizer_of(tensor):
    rec = of(tensor)
    return None if rec is None else rec[0]


def indices_of(tensor):
    rec = of(tensor)
    return None if rec is None else rec[1]
