"""ctypes binding of libtq_hip.so (include/tq_hip.h) + the tensor-level backend the
quantization classes call.

There is NO CPU implementation here or anywhere else in this package: if the shared library is
missing, or a tensor is not resident on a ROCm device, the call raises.  (The test-suite swaps
`backend()` for an oracle-backed double to exercise the host-side state machines on a box
without a GPU; that double lives under tests/, not here.)
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TQ_LIB_PATH') or os.path.join(os.path.dirname(_HERE), 'lib', 'libtq_hip.so')   # TQ_LIB_PATH: A/B of two builds on one box

TQ_F32, TQ_BF16, TQ_F16 = 0, 1, 2
IDX_NONE, IDX_F32, IDX_I8, IDX_U8, IDX_I16, IDX_I32, IDX_I8_M128 = range(7)
ACT_NONE, ACT_RELU, ACT_GELU, ACT_TANH = range(4)
EST_CURRENT, EST_ALL, EST_RUNNING = 0, 1, 2
ADA_SIGMOID, ADA_HARD_SIGMOID, ADA_SIGMOID_TEMP = 0, 1, 2

_DTYPES = {torch.float32: TQ_F32, torch.bfloat16: TQ_BF16, torch.float16: TQ_F16}
_IDX_DTYPES = {torch.float32: IDX_F32, torch.int8: IDX_I8, torch.uint8: IDX_U8,
               torch.int16: IDX_I16, torch.int32: IDX_I32}


class TQError(RuntimeError):
    pass


class tq_quantizer(C.Structure):
    _fields_ = [('delta', C.c_void_p), ('zero_float', C.c_void_p), ('signed_flag', C.c_void_p),
                ('n_bits', C.c_int32), ('symmetric', C.c_int32), ('log_domain', C.c_int32),
                ('eps', C.c_float), ('n_params', C.c_uint64), ('inner', C.c_uint64)]


class tq_fq_item(C.Structure):          # one tensor of tq_fake_quant_multi_fwd
    _fields_ = [('x', C.c_void_p), ('y', C.c_void_p), ('n', C.c_uint64), ('q', tq_quantizer)]


class tq_quantizer_f64(C.Structure):
    _fields_ = [('delta', C.c_void_p), ('zero_float', C.c_void_p), ('signed_flag', C.c_void_p),
                ('n_bits', C.c_int32), ('symmetric', C.c_int32), ('log_domain', C.c_int32),
                ('reserved', C.c_int32), ('eps', C.c_double), ('n_params', C.c_uint64), ('inner', C.c_uint64)]


_u64, _vp, _int, _sz, _f, _d = C.c_uint64, C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_double
_QP = C.POINTER(tq_quantizer)


class tq_ffn_stage(C.Structure):
    """include/tq_hip.h: one block of tq_ffn_chain_i8_nonorm_fwd"""
    _fields_ = [('w1_idx', C.c_void_p), ('w1_rowsum', C.c_void_p), ('bias1', C.c_void_p), ('w1_delta', C.c_void_p),
                ('w1_n_params', C.c_uint64), ('w1_eps', C.c_float), ('q_mid', _QP),
                ('w2_idx', C.c_void_p), ('w2_rowsum', C.c_void_p), ('bias2', C.c_void_p), ('w2_delta', C.c_void_p),
                ('w2_n_params', C.c_uint64), ('w2_eps', C.c_float),
                ('nn_weight', C.c_void_p), ('nn_bias', C.c_void_p), ('q_dense', _QP), ('q_sum', _QP), ('q_out', _QP)]
_QPD = C.POINTER(tq_quantizer_f64)

# name -> (restype, argtypes); must list every symbol include/tq_hip.h declares

ABI_VERSION = 2          # == TQ_ABI_VERSION of include/tq_hip.h (tests/test_abi.py compares the header, the library and this)

SIGNATURES = {
    'tq_fake_quant_fwd_f64': (_int, [_vp, _vp, _vp, _u64, _QPD, _vp]),
    'tq_fake_quant_bwd_f64_workspace_bytes': (_sz, [_u64, _u64, _u64]),
    'tq_fake_quant_bwd_f64': (_int, [_vp, _vp, _vp, _vp, _vp, _u64, _QPD, _vp, _sz, _vp]),
    'tq_minmax_f64_workspace_bytes': (_sz, [_u64, _u64, _u64]),
    'tq_minmax_f64': (_int, [_vp, _u64, _u64, _u64, _vp, _vp, _vp, _sz, _vp]),
    'tq_range_update_f64': (_int, [_int, _vp, _vp, _vp, _vp, _u64, _int, _d, _u64, _vp, _vp]),
    'tq_axis_ranges_f64': (_int, [_vp, _vp, _vp, _u64, _int, _vp]),
    'tq_set_range_asym_f64': (_int, [_vp, _vp, _u64, _int, _d, _int, _vp, _vp, _vp]),
    'tq_set_range_sym_f64': (_int, [_vp, _vp, _u64, _int, _d, _int, _vp, _vp, _vp]),
    'tq_mse_candidates_f64': (_int, [_vp, _u64, _u64, _vp, _u64, _int, _vp, _vp]),
    'tq_abi_version': (_int, []),
    'tq_last_error': (C.c_char_p, []),
    'tq_fake_quant_fwd': (_int, [_vp, _vp, _vp, _int, _u64, _int, _QP, _vp]),
    'tq_fake_quant_multi_fwd': (_int, [C.POINTER(tq_fq_item), C.c_uint32, _int, _vp]),
    'tq_affine_fake_quant_fwd': (_int, [_vp, _vp, _vp, _vp, _vp, _u64, _u64, _int, _QP, _vp]),
    'tq_residual_layernorm_quant_fwd': (_int, [_vp, _vp, _vp, _vp, _u64, _u64, _int, _QP, _QP, _vp, _vp, _f, _QP, _vp]),
    'tq_residual_nonorm_quant_fwd': (_int, [_vp, _vp, _vp, _vp, _u64, _u64, _int, _QP, _QP, _vp, _vp, _QP, _vp]),
    'tq_embeddings_layernorm_quant_fwd': (_int, [_vp, _u64, _vp, _vp, _u64, _vp, _vp, _u64, _vp, _vp, _vp, _u64, _u64, _QP, _QP,
                                                _vp, _vp, C.c_float, _QP, _vp, _vp]),
    'tq_attention_i8_fwd': (_int, [_vp, _vp, _vp, _vp, _vp, _u64, _u64, _u64, _u64, _u64, _vp, _f, _QP, _QP, _QP, _QP, _QP, _QP, _vp]),
    'tq_attention_i8_strided_fwd': (_int, [_vp, _vp, _vp, _vp, _vp, _u64, _u64, _u64, _u64, _u64, _u64, _vp, _f, _QP, _QP, _QP, _QP,
                                           _QP, _QP, _vp]),
    'tq_linear_i8_grouped_fwd': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _u64, _u64, _u64, _vp, _vp, _int, _f, _vp, _f, _int,
                                        _u64, C.POINTER(_QP), _vp]),
    'tq_scores_softmax_quant_fwd': (_int, [_vp, _vp, _u64, _u64, _vp, _u64, _f, _QP, _QP, _vp]),
    'tq_rowsum_i8': (_int, [_vp, _vp, _u64, _u64, _vp]),
    'tq_ffn_i8_nonorm_fwd': (_int, [_vp, _vp, _vp, _int, _f, _vp, _vp, _vp, _vp, _u64, _f, _QP, _vp, _vp, _vp, _vp, _u64, _f,
                                    _vp, _vp, _vp, _QP, _QP, _QP, _vp, _vp, _int, _u64, _u64, _u64, _u64, _vp]),
    'tq_linear_i8_nonorm_fwd': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _u64, _u64, _u64, _vp, _vp, _int, _f,
                                       _vp, _u64, _f, _QP, _QP, _QP, _vp]),
    'tq_ffn_chain_i8_nonorm_fwd': (_int, [_vp, _vp, _vp, _int, _f, _vp, _vp, _u64, _vp, _vp, _int, _u64, _u64, _u64, _u64, _vp]),
    'tq_linear_i8_nonorm_grouped_fwd': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _u64, _u64, _u64, _vp, _vp, _int, _f, _vp,
                                               _f, _u64, C.POINTER(_QP), C.POINTER(_QP), _vp]),
    'tq_linear_i8_fwd': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _u64, _u64, _u64, _vp, _vp, _int, _f, _vp, _u64, _f,
                                _int, _QP, _vp]),
    'tq_linear_i8_stair_fwd': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _u64, _u64, _u64, _vp, _vp, _int, _f, _vp, _u64, _f,
                                      _int, _QP, _vp, C.c_uint32, _vp]),
    'tq_act_stair_bytes': (_sz, [C.c_uint32]),
    'tq_act_stair_build': (_int, [_int, _QP, C.c_uint32, _vp, _sz, _vp]),
    'tq_fake_quant_bwd_workspace_bytes': (_sz, [_u64]),
    'tq_fake_quant_bwd_params_workspace_bytes': (_sz, [_u64, _u64, _u64]),
    'tq_fake_quant_bwd': (_int, [_vp, _vp, _vp, _vp, _vp, _u64, _int, _QP, _vp, _sz, _vp]),
    'tq_minmax_workspace_bytes': (_sz, [_u64, _u64, _u64]),
    'tq_minmax': (_int, [_vp, _u64, _int, _u64, _u64, _vp, _vp, _vp, _sz, _vp]),
    'tq_calibrate_workspace_bytes': (_sz, [_u64, _u64, _u64]),
    'tq_calibrate_minmax': (_int, [_vp, _u64, _int, _u64, _u64, _int, _vp, _vp, _vp, _vp, _d, _u64, _vp, _int,
                                   _int, _f, _int, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'tq_calibrate_tensor': (_int, [_vp, _u64, _int, _int, _vp, _vp, _vp, _vp, _d, _int, _int, _f, _int, _vp, _vp, _vp,
                                   _vp, _vp, _sz, _vp, _vp]),
    'tq_calibrate_stats': (_int, [_vp, _u64, _int, _u64, _u64, _vp, _vp, _sz, _vp, _vp]),
    'tq_calibrate_apply': (_int, [_vp, _vp, _u64, _int, _u64, _u64, _int, _vp, _vp, _vp, _vp, _d, _u64, _vp, _int, _int, _f,
                                  _int, _vp, _vp, _vp, _vp, _vp]),
    'tq_mailbox_bytes': (_sz, []),
    'tq_mailbox_max_floats': (_sz, []),
    'tq_mailbox_handle_bytes': (_sz, []),
    'tq_mailbox_alloc': (_int, [C.POINTER(_vp), _vp]),
    'tq_mailbox_open': (_int, [_vp, C.POINTER(_vp)]),
    'tq_mailbox_close': (_int, [_vp]),
    'tq_mailbox_free': (_int, [_vp]),
    'tq_mailbox_allreduce_max': (_int, [_vp, _u64, _vp, _vp, C.c_uint32, C.c_uint32, _vp, C.c_uint32, _vp]),
    'tq_calibrate_minmax_mailbox': (_int, [_vp, _u64, _int, _u64, _u64, _int, _vp, _vp, _vp, _vp, _d, _u64, _vp, _int, _int, _f,
                                           _int, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, C.c_uint32, C.c_uint32, _vp,
                                           C.c_uint32, _vp]),
    'tq_order_stats_workspace_bytes': (_sz, [_u64, C.c_uint32]),
    'tq_order_stats': (_int, [_vp, _u64, _u64, _int, C.POINTER(_u64), C.c_uint32, _vp, _vp, _sz, _vp]),
    'tq_comm_unique_id_bytes': (_sz, []),
    'tq_comm_load': (_int, [C.c_char_p]),
    'tq_comm_version': (_int, []),
    'tq_comm_get_unique_id': (_int, [_vp]),
    'tq_comm_init': (_int, [_vp, _int, _int, C.POINTER(_vp)]),
    'tq_comm_destroy': (_int, [_vp]),
    'tq_comm_abort': (_int, [_vp]),
    'tq_comm_rank_world': (_int, [_vp, C.POINTER(_int), C.POINTER(_int)]),
    'tq_comm_allreduce': (_int, [_vp, _vp, _u64, _int, _int, _vp]),
    'tq_comm_broadcast': (_int, [_vp, _vp, _u64, _int, _int, _vp]),
    'tq_calibrate_minmax_rccl': (_int, [_vp, _u64, _int, _u64, _u64, _int, _vp, _vp, _vp, _vp, _d, _u64, _vp, _int, _int, _f,
                                        _int, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    'tq_range_update': (_int, [_int, _vp, _vp, _vp, _vp, _u64, _int, _d, _u64, _vp, _vp]),
    'tq_axis_ranges': (_int, [_vp, _vp, _vp, _u64, _int, _vp]),
    'tq_set_range_asym': (_int, [_vp, _vp, _u64, _int, _f, _int, _vp, _vp, _vp]),
    'tq_set_range_sym': (_int, [_vp, _vp, _u64, _int, _f, _int, _vp, _vp, _vp]),
    'tq_mse_workspace_bytes': (_sz, [_u64, _u64, _u64]),
    'tq_mse_candidates': (_int, [_vp, _u64, _u64, _int, _vp, _u64, _vp, _vp, _sz, _vp]),
    'tq_mse_candidates_grouped': (_int, [_vp, _u64, _u64, _u64, _int, _vp, _u64, _vp, _vp, _sz, _vp]),
    'tq_mse_ordered_workspace_bytes': (_sz, [_u64, _u64, _u64]),
    'tq_mse_candidates_ordered': (_int, [_vp, _u64, _u64, _int, _vp, _u64, _int, _vp, _vp, _vp, _sz, _vp]),
    'tq_xent_candidates': (_int, [_vp, _u64, _u64, _vp, _u64, _vp, _vp]),
    'tq_argmin_select': (_int, [_vp, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp]),
    'tq_adaround_fwd': (_int, [_vp, _vp, _vp, _u64, _QP, _int, _int, _f, _vp]),
    'tq_adaround_init_alpha': (_int, [_vp, _vp, _u64, _QP, _int, _f, _vp]),
    'tq_adaround_bwd': (_int, [_vp, _vp, _vp, _vp, _u64, _QP, _int, _f, _vp]),
    'tq_adaround_bwd_adam_sched': (_int, [_vp, _vp, _vp, _vp, _vp, _u64, _QP, _int, _f, _vp, _f, _f, _f, _f, _vp]),
    'tq_adaround_bwd_adam': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _u64, _QP, _int, _f, _f, _f, _f,
                                    _f, _f, _f, _int, _vp]),
    'tq_adaround_reg': (_int, [_vp, _u64, _int, _f, _f, _f, _vp, _vp, _sz, _vp]),
    'tq_reduce_workspace_bytes': (_sz, [_u64]),
    'tq_recon_loss': (_int, [_vp, _vp, _u64, _u64, _u64, _vp, _vp, _sz, _vp]),
}

_lib = None


def load_library(path=None):
    """dlopen libtq_hip.so and declare every prototype.  Raises if the library is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise TQError(
            f'{p} not found: build it with `python transformer-quantization_amd/build.py` '
            '(hipcc --offload-arch=gfx950).  There is no CPU fallback.')
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.tq_abi_version() != ABI_VERSION:
        raise TQError('libtq_hip.so ABI version mismatch: the library says %d, this binding is written for %d -- rebuild '
                      '(python transformer-quantization_amd/build.py)' % (lib.tq_abi_version(), ABI_VERSION))
    if path is None:
        _lib = lib
    return lib


_fastcall_mod, _fastcall_tried = None, False


def fastcall():
    """The CPython stub over the C ABI (csrc_py/tq_fastcall.c -> lib/_tq_fastcall*.so: plain C, no torch headers) or None.
    It calls the SAME entry points of the SAME mapped libtq_hip.so through their raw addresses, without ctypes' argument
    marshalling (~2 us per call).  Optional: when it is not built, or with TQ_FASTCALL=0, every call goes through ctypes."""
    global _fastcall_mod, _fastcall_tried
    if _fastcall_tried:
        return _fastcall_mod
    _fastcall_tried = True
    if os.environ.get('TQ_FASTCALL', '1') == '0':
        return None
    import glob
    import importlib.machinery
    import importlib.util
    for path in sorted(glob.glob(os.path.join(os.path.dirname(LIB_PATH), '_tq_fastcall*.so'))):
        try:
            loader = importlib.machinery.ExtensionFileLoader('_tq_fastcall', path)
            spec = importlib.util.spec_from_loader('_tq_fastcall', loader)
            mod = importlib.util.module_from_spec(spec)
            loader.exec_module(mod)
            _fastcall_mod = mod
            break
        except Exception:       # noqa: BLE001 -- built for another interpreter: ctypes it is
            continue
    return _fastcall_mod


def entry_address(fn):
    """Raw address of a ctypes function of the loaded library (what `fastcall()` calls through)."""
    return C.cast(fn, C.c_void_p).value


def _check(rc, lib):
    if rc != 0:
        raise TQError(f'libtq_hip: {lib.tq_last_error().decode()} (code {rc})')


def _ptr(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """hipStream_t of torch's current stream on the current device (the raw getter is ~10x cheaper than
    building a torch.cuda.Stream object per call; launch-bound calibration makes 161 such calls per batch)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _need_device(t, what):
    if not t.is_cuda:
        raise TQError(f'{what}: tensor lives on {t.device}; the fake-quant path only runs on a '
                      'ROCm device (no CPU fallback)')


def _need_f32(what, *tensors):
    """The AdaRound kernels are fp32-only (weights, alpha, Adam moments): refuse anything else instead of
    reinterpreting its bytes."""
    for t in tensors:
        if t is not None and t.dtype != torch.float32:
            raise TQError(f'{what}: fp32 tensors required, got {t.dtype} (call .float() on the layer first)')


def _dtype_code(t, what):
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        hint = ' -- float64 (--double) runs the layered quantizer path only' if t.dtype == torch.float64 else ''
        raise TQError(f'{what}: dtype {t.dtype} not supported (fp32 / bf16 / fp16){hint}') from None


def _device_guarded(fn):
    """Run a backend method on the device its tensor operands live on: workspaces, ticket words, the raw stream
    handle and every output allocation follow torch's CURRENT device, so a tensor on cuda:1 while cuda:0 is current
    (nn.DataParallel replicas, `keep_gpu` caches, a model moved to another GPU) would be launched on the wrong
    device's stream against foreign pointers.  ATen ops switch device automatically; so does this.  Operands on
    different devices raise instead of faulting."""
    import functools

    _Tensor, _cur = torch.Tensor, torch.cuda.current_device

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        # hot path (launch-bound calibration makes 263 such calls per batch): one pass over the positional operands
        idx = -1
        for a in args:
            if isinstance(a, _Tensor):
                i = a.get_device()              # device index, -1 for host tensors: one call instead of is_cuda + device.index
                if i < 0:
                    continue
                if idx < 0:
                    idx = i
                elif i != idx:
                    raise TQError(f'{fn.__name__}: operands live on different devices (cuda:{idx} and cuda:{i})')
        if idx < 0:
            # no top-level tensor operand: quantizer tuples (delta, zero_float, ...) carry the device (act_stair)
            for a in args:
                if isinstance(a, (tuple, list)):
                    for b in a:
                        if isinstance(b, _Tensor) and b.get_device() >= 0:
                            idx = b.get_device()
                            break
                    if idx >= 0:
                        break
        if idx < 0 or idx == _cur():
            return fn(self, *args, **kwargs)
        with torch.cuda.device(idx):
            return fn(self, *args, **kwargs)
    return wrapper


class HipBackend:
    """Tensor-level wrappers; one instance per process.  All work goes to torch's current stream on the device of
    the operands (every public method is wrapped by `_device_guarded`, see the bottom of the class)."""

    name = 'hip'

    def __init__(self):
        self.lib = load_library()
        self._ws = {}
        self._qdesc_cache = {}
        self._ws_bytes = {}
        self._counters = {}      # zeroed ticket words of tq_calibrate_tensor, one per (device, stream)
        self._calib_tensor_addr = None      # raw address of tq_calibrate_tensor for the CPython stub

    # -- helpers -------------------------------------------------------------------------
    def _workspace(self, device, nbytes):
        key = (device.index, _stream())
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    def _calib_ws_bytes(self, n, n_params, inner):
        """tq_calibrate_workspace_bytes, memoised (a pure function of its arguments; one C call less per calibrating
        forward of every quantizer)."""
        key = (n, n_params, inner)
        v = self._ws_bytes.get(key)
        if v is None:
            if len(self._ws_bytes) > 4096:
                self._ws_bytes.clear()
            v = self._ws_bytes[key] = self.lib.tq_calibrate_workspace_bytes(n, n_params, inner)
        return v

    def to_device_f32(self, v, like=None):
        """python scalar / numpy / CPU tensor -> fp32 tensor on the active ROCm device."""
        if torch.is_tensor(v):
            if v.is_cuda:
                return v if v.dtype == torch.float32 else v.float()
            dev = like.device if (like is not None and like.is_cuda) else torch.device(
                'cuda', torch.cuda.current_device())
            return v.detach().to(device=dev, dtype=torch.float32)
        dev = like.device if (like is not None and like.is_cuda) else torch.device(
            'cuda', torch.cuda.current_device())
        return torch.tensor(v, dtype=torch.float64).float().to(dev)

    def _qdesc(self, delta, zero_float, signed, n_bits, symmetric, log_domain, eps, n_params, inner):
        """The C descriptor of a quantizer.  Building a 9-field ctypes struct is ~4 us of host time -- a quarter of a
        launch-bound fixed-range call -- so descriptors are memoised by VALUE (pointers and scalars: the key
        determines every field, nothing can go stale; a struct is never written after construction)."""
        key = (None if delta is None else delta.data_ptr(), None if zero_float is None else zero_float.data_ptr(),
               None if signed is None else signed.data_ptr(), n_bits, symmetric, log_domain, eps, n_params, inner)
        cache = self._qdesc_cache
        d = cache.get(key)
        if d is None:
            if len(cache) > 8192:        # calibration rebinds the range tensors every step: bounded, not LRU
                cache.clear()
            d = cache[key] = tq_quantizer(key[0], key[1], key[2], int(n_bits), int(bool(symmetric)),
                                          int(bool(log_domain)), float(eps), int(n_params), int(inner))
        return d

    # -- FP64 (`--double`) ------------------------------------------------------------------
    @staticmethod
    def _as_f64(t, like):
        """Range buffer -> contiguous float64 device tensor (exact widening of an fp32 buffer: torch's own type
        promotion in `x / scale`); None stays None."""
        if t is None:
            return None
        t = t.detach()
        if t.dtype != torch.float64 or t.device != like.device:
            t = t.to(device=like.device, dtype=torch.float64)
        return t.contiguous()

    def _qdesc_f64(self, delta, zero_float, signed, n_bits, symmetric, log_domain, eps, n_params, inner):
        return tq_quantizer_f64(_ptr(delta), _ptr(zero_float), _ptr(signed), int(n_bits), int(bool(symmetric)),
                                int(bool(log_domain)), 0, float(eps), int(n_params), int(inner))

    def _fake_quant_f64(self, x, delta, zero_float, signed, n_bits, symmetric, log_domain, eps, n_params, inner,
                        want_y, want_idx):
        x = x.contiguous()
        d, z = self._as_f64(delta, x), self._as_f64(zero_float, x)
        y = torch.empty_like(x) if want_y else None
        idx = torch.empty_like(x) if want_idx else None
        q = self._qdesc_f64(d, z, signed, n_bits, symmetric, log_domain, eps, n_params, inner)
        rc = self.lib.tq_fake_quant_fwd_f64(_ptr(x), _ptr(y), _ptr(idx), x.numel(), C.byref(q), _stream())
        _check(rc, self.lib)
        return y, idx

    # -- K1/K2/K3 ------------------------------------------------------------------------
    def fake_quant(self, x, delta, zero_float, signed, n_bits, symmetric, log_domain, eps,
                   n_params, inner, want_y=True, idx_dtype=None):
        _need_device(x, 'fake_quant')
        if x.dtype == torch.float64:
            if idx_dtype not in (None, torch.float32, torch.float64):
                raise TQError('fake_quant: float64 tensors emit float64 indices only')
            return self._fake_quant_f64(x, delta, zero_float, signed, n_bits, symmetric, log_domain, eps, n_params,
                                        inner, want_y, idx_dtype is not None)
        x = x.contiguous()
        y = torch.empty_like(x) if want_y else None
        idx = torch.empty(x.shape, dtype=idx_dtype, device=x.device) if idx_dtype is not None else None
        q = self._qdesc(delta, zero_float, signed, n_bits, symmetric, log_domain, eps, n_params, inner)
        rc = self.lib.tq_fake_quant_fwd(_ptr(x), _ptr(y), _ptr(idx),
                                        _IDX_DTYPES[idx_dtype] if idx_dtype is not None else IDX_NONE,
                                        x.numel(), _dtype_code(x, 'fake_quant'), C.byref(q), _stream())
        _check(rc, self.lib)
        return y, idx

    def fake_quant_multi_plan(self, items):
        """(C item table, outputs, inputs kept alive, dtype code) for `fake_quant_multi_launch`: build once, launch many
        times (the outputs are overwritten in place)."""
        arr = (tq_fq_item * max(len(items), 1))()
        ys, keep = [], []
        dtype = items[0][0].dtype if items else torch.float32
        for k, (x, delta, zf, signed, n_bits, symmetric, log_domain, eps, n_params, inner) in enumerate(items):
            _need_device(x, 'fake_quant_multi')
            if x.dtype != dtype:
                raise TQError('fake_quant_multi: all tensors must share one dtype')
            x = x.contiguous()
            y = torch.empty_like(x)
            keep.append((x, delta, zf, signed))
            ys.append(y)
            arr[k].x, arr[k].y, arr[k].n = x.data_ptr(), y.data_ptr(), x.numel()
            arr[k].q = tq_quantizer(_ptr(delta), _ptr(zf), _ptr(signed), int(n_bits), int(bool(symmetric)),
                                    int(bool(log_domain)), float(eps), int(n_params), int(inner))
        return arr, ys, keep, (_dtype_code(items[0][0], 'fake_quant_multi') if items else 0)

    def fake_quant_multi_launch(self, plan):
        arr, ys, _, code = plan
        if ys:
            _check(self.lib.tq_fake_quant_multi_fwd(arr, len(ys), code, _stream()), self.lib)
        return ys

    def fake_quant_multi(self, items):
        """Fake-quantize many independent tensors of ONE dtype in one launch (40 per launch, more are split).
        items: [(x, delta, zero_float, signed, n_bits, symmetric, log_domain, eps, n_params, inner), ...] -> [y, ...];
        bit-identical to `fake_quant` on each item.  Per-row parameters need rows of whole 16-byte vectors."""
        return self.fake_quant_multi_launch(self.fake_quant_multi_plan(items)) if items else []

    def fixed_quant_plan(self, delta, zero_float, signed, n_bits, symmetric, log_domain, eps):
        """Launch plan of a fixed-range per-tensor quantizer for QuantizationManager's fast path: the memoised C
        descriptor as a ctypes reference plus the entry point, so that a fixed-range call is torch.empty_like + ONE
        foreign call (the generic route through QuantizerBase.forward -> fake_quant costs ~6 us of Python per call,
        x 161 / 1333 quantizer calls per BERT-base / MobileBERT forward)."""
        d = self._qdesc(delta, zero_float, signed, n_bits, symmetric, log_domain, eps, 1, 1)
        fc = fastcall()
        if fc is not None:      # raw addresses for the CPython stub; `d` (kept in the plan) owns the descriptor memory
            return fc.fake_quant_fwd, (entry_address(self.lib.tq_fake_quant_fwd), C.addressof(d)), d
        return self.lib.tq_fake_quant_fwd, C.byref(d), d

    def affine_fake_quant(self, x, w, b, delta, zero_float, signed, n_bits, symmetric, log_domain, eps, want_idx=False):
        """y = Q(x * w + b) (w, b fp32 [d] over the last axis), per-tensor output quantizer.  want_idx (asymmetric
        <= 8-bit quantizer): also int8(index - 128) of y, from the same launch -> (y, idx)."""
        _need_device(x, 'affine_fake_quant')
        x = x.contiguous()
        y = torch.empty_like(x)
        idx = torch.empty(x.shape, dtype=torch.int8, device=x.device) if want_idx else None
        q = self._qdesc(delta, zero_float, signed, n_bits, symmetric, log_domain, eps, 1, 1)
        rc = self.lib.tq_affine_fake_quant_fwd(_ptr(x), _ptr(w.detach().float().contiguous()),
                                               _ptr(b.detach().float().contiguous()), _ptr(y), _ptr(idx), x.numel(),
                                               x.shape[-1], _dtype_code(x, 'affine_fake_quant'),
                                               C.byref(q), _stream())
        _check(rc, self.lib)
        return (y, idx) if want_idx else y

    def residual_layernorm_quant(self, dense_out, residual, q_dense, q_sum, ln_weight, ln_bias, ln_eps, q_out,
                                 want_idx=False):
        """y = Q_out(LN(Q_sum(Q_dense(dense_out) + residual))); each q_* is None or the 7-tuple
        (delta, zero_float, signed, n_bits, symmetric, log_domain, eps) of a per-tensor quantizer.
        ln_eps=None selects MobileBERT's NoNorm (u * w + b) instead of LayerNorm."""
        _need_device(dense_out, 'residual_layernorm_quant')
        a, r = dense_out.contiguous(), residual.contiguous().to(dense_out.dtype)
        y = torch.empty_like(a)
        idx = torch.empty(a.shape, dtype=torch.int8, device=a.device) if want_idx else None
        d = a.shape[-1]
        descs = [None if q is None else self._qdesc(*q, 1, 1) for q in (q_dense, q_sum, q_out)]
        refs = [None if dsc is None else C.byref(dsc) for dsc in descs]
        w32, b32 = ln_weight.detach().float().contiguous(), ln_bias.detach().float().contiguous()
        if ln_eps is None:          # NoNorm: element-wise affine, no statistics
            rc = self.lib.tq_residual_nonorm_quant_fwd(
                _ptr(a), _ptr(r), _ptr(y), _ptr(idx), a.numel() // d, d, _dtype_code(a, 'residual_layernorm_quant'),
                refs[0], refs[1], _ptr(w32), _ptr(b32), refs[2], _stream())
        else:
            rc = self.lib.tq_residual_layernorm_quant_fwd(
                _ptr(a), _ptr(r), _ptr(y), _ptr(idx), a.numel() // d, d, _dtype_code(a, 'residual_layernorm_quant'),
                refs[0], refs[1], _ptr(w32), _ptr(b32), float(ln_eps), refs[2], _stream())
        _check(rc, self.lib)
        return (y, idx) if want_idx else y

    def embeddings_layernorm_quant(self, word, word_ids, type_tab, type_ids, pos_tab, pos_ids, q_sum1, q_sum2, ln_weight,
                                   ln_bias, ln_eps, q_out, want_idx=False):
        """y [rows, d] = Q_out(LN(Q_sum2(Q_sum1(word[word_ids] + type[type_ids]) + pos[pos_ids]))); tables fp32 [n, d]
        (already fake-quantized), ids int64 [rows]; each q_* None or a per-tensor 7-tuple.  -> y (, int8 indices).
        An id outside its table makes its output row NaN and raises IndexError (what torch's CPU F.embedding raises) at the
        next call of this method or of `raise_deferred` -- the launch itself is asynchronous."""
        _need_device(word, 'embeddings_layernorm_quant')
        self.raise_deferred()
        _need_f32('embeddings_layernorm_quant', word, type_tab, pos_tab)
        tabs = [t.detach().contiguous() for t in (word, type_tab, pos_tab)]
        ids = [i.reshape(-1).contiguous() for i in (word_ids, type_ids, pos_ids)]
        rows, d = ids[0].numel(), tabs[0].shape[-1]
        if any(i.numel() != rows or i.dtype != torch.int64 for i in ids) or any(t.shape[-1] != d for t in tabs):
            raise TQError('embeddings_layernorm_quant: ids must be int64 of one length, tables of one width')
        y = torch.empty((rows, d), dtype=torch.float32, device=word.device)
        idx = torch.empty((rows, d), dtype=torch.int8, device=word.device) if want_idx else None
        descs = [None if q is None else self._qdesc(*q, 1, 1) for q in (q_sum1, q_sum2, q_out)]
        refs = [None if dsc is None else C.byref(dsc) for dsc in descs]
        w32, b32 = ln_weight.detach().float().contiguous(), ln_bias.detach().float().contiguous()
        rc = self.lib.tq_embeddings_layernorm_quant_fwd(
            _ptr(tabs[0]), tabs[0].shape[0], _ptr(ids[0]), _ptr(tabs[1]), tabs[1].shape[0], _ptr(ids[1]), _ptr(tabs[2]),
            tabs[2].shape[0], _ptr(ids[2]), _ptr(y), _ptr(idx), rows, d, refs[0], refs[1], _ptr(w32), _ptr(b32), float(ln_eps),
            refs[2], self._bad_ids_flag().data_ptr(), _stream())
        _check(rc, self.lib)
        return (y, idx) if want_idx else y

    def _bad_ids_flag(self):
        """4 bytes of pinned host memory (device-visible at the same address): kernels raise it, the host reads it without
        synchronising.  One per process (one process per GPU)."""
        f = self.__dict__.get('_bad_ids')
        if f is None:
            f = self.__dict__['_bad_ids'] = torch.zeros(1, dtype=torch.int32).pin_memory()
        return f

    def raise_deferred(self, sync=False):
        """Errors that launches already queued could only report asynchronously (an embedding id outside its table):
        raised here, once.  sync=True waits for the device first (use after the last forward of an evaluation)."""
        f = self.__dict__.get('_bad_ids')
        if f is None:
            return
        if sync and not torch.cuda.is_current_stream_capturing():
            torch.cuda.synchronize()
        if int(f[0]) != 0:
            f.zero_()
            raise IndexError('index out of range in self (an embedding id outside its table reached '
                             'tq_embeddings_layernorm_quant_fwd; the rows it produced are NaN)')

    def attention_i8(self, q_idx, k_idx, v_idx, num_heads, mask, denom, q_q, q_k, q_v, q_scores, q_probs, q_ctx,
                     want_idx=False):
        """Quantized attention core on int8 indices [B, T, H * 64] (contiguous tensors, or column blocks of stacked
        buffers: Q | K | V of one [B, T, 3 * H * 64] buffer, or Q | K of one buffer and V of another); every q_* is a
        per-tensor 7-tuple (q_scores / q_ctx may be None).  -> ctx fp32 [B, T, H * 64] (, int8 indices of ctx)."""
        _need_device(q_idx, 'attention_i8')
        B, T, D = q_idx.shape
        rows = lambda t: t.stride(2) == 1 and t.stride(0) == T * t.stride(1) and t.stride(1) % 16 == 0
        if not (rows(q_idx) and rows(k_idx) and rows(v_idx) and k_idx.stride() == q_idx.stride()):
            q_idx, k_idx, v_idx = q_idx.contiguous(), k_idx.contiguous(), v_idx.contiguous()
        ctx = torch.empty((B, T, D), dtype=torch.float32, device=q_idx.device)
        ctx_idx = torch.empty((B, T, D), dtype=torch.int8, device=q_idx.device) if want_idx else None
        descs = [None if q is None else self._qdesc(*q, 1, 1) for q in (q_q, q_k, q_v, q_scores, q_probs, q_ctx)]
        refs = [None if dsc is None else C.byref(dsc) for dsc in descs]
        rc = self.lib.tq_attention_i8_strided_fwd(_ptr(q_idx), _ptr(k_idx), _ptr(v_idx), _ptr(ctx), _ptr(ctx_idx), B, T,
                                                  num_heads, D // num_heads, q_idx.stride(1), v_idx.stride(1), _ptr(mask),
                                                  float(denom), *refs, _stream())
        _check(rc, self.lib)
        return (ctx, ctx_idx) if want_idx else ctx

    def linear_i8_grouped(self, x_idx, w_idx, w_rowsum, bias, x_q, w_delta_rows, w_eps, activation, q_outs,
                          want_y=False, want_idx=True, out_dtype=torch.float32):
        """len(q_outs) Linears sharing x_idx as one launch: w_idx / w_rowsum / bias / w_delta_rows stacked along
        N, q_outs[g] the per-tensor 7-tuple of group g's output quantizer.  -> (y | None, y_idx | None)."""
        K = x_idx.shape[-1]
        M = x_idx.numel() // K
        N = w_idx.shape[0]
        y = torch.empty(x_idx.shape[:-1] + (N,), dtype=out_dtype, device=x_idx.device) if want_y else None
        y_idx = torch.empty(x_idx.shape[:-1] + (N,), dtype=torch.int8, device=x_idx.device) if want_idx else None
        descs = [self._qdesc(*q, 1, 1) for q in q_outs]
        arr = (C.POINTER(tq_quantizer) * len(descs))(*[C.pointer(d) for d in descs])
        rc = self.lib.tq_linear_i8_grouped_fwd(
            _ptr(x_idx), _ptr(w_idx), _ptr(w_rowsum), _ptr(bias), _ptr(y), _ptr(y_idx), _DTYPES[out_dtype], M, N, K,
            _ptr(x_q[0]), _ptr(x_q[1]), int(x_q[2]), float(x_q[3]), _ptr(w_delta_rows), float(w_eps), int(activation),
            len(descs), C.cast(arr, C.POINTER(_QP)), _stream())
        _check(rc, self.lib)
        return y, y_idx

    def scores_softmax_quant(self, scores, mask, rows_per_mask, denom, q_scores, q_probs):
        """probs = Q_probs(softmax(Q_scores(scores) / denom + mask)); q_* None or per-tensor 7-tuples."""
        _need_device(scores, 'scores_softmax_quant')
        s = scores.contiguous()
        y = torch.empty_like(s)
        cols = s.shape[-1]
        descs = [None if q is None else self._qdesc(*q, 1, 1) for q in (q_scores, q_probs)]
        refs = [None if dsc is None else C.byref(dsc) for dsc in descs]
        rc = self.lib.tq_scores_softmax_quant_fwd(_ptr(s), _ptr(y), s.numel() // cols, cols, _ptr(mask),
                                                  int(rows_per_mask), float(denom), refs[0], refs[1], _stream())
        _check(rc, self.lib)
        return y

    def fake_quant_int8(self, x, delta, zero_float, n_bits, eps):
        """Per-tensor asymmetric fake-quant that also emits int8(index - 128): one read, two writes."""
        _need_device(x, 'fake_quant_int8')
        x = x.contiguous()
        y = torch.empty_like(x)
        idx = torch.empty(x.shape, dtype=torch.int8, device=x.device)
        q = self._qdesc(delta, zero_float, None, n_bits, False, False, eps, 1, 1)
        rc = self.lib.tq_fake_quant_fwd(_ptr(x), _ptr(y), _ptr(idx), IDX_I8_M128, x.numel(),
                                        _dtype_code(x, 'fake_quant_int8'), C.byref(q), _stream())
        _check(rc, self.lib)
        return y, idx

    def quantize_to_int8(self, x, delta, zero_float, signed, n_bits, symmetric, log_domain, eps, n_params,
                         inner, minus_128):
        """int8 grid indices of x (minus 128 for unsigned activation grids)."""
        _need_device(x, 'quantize_to_int8')
        x = x.contiguous()
        idx = torch.empty(x.shape, dtype=torch.int8, device=x.device)
        q = self._qdesc(delta, zero_float, signed, n_bits, symmetric, log_domain, eps, n_params, inner)
        rc = self.lib.tq_fake_quant_fwd(_ptr(x), None, _ptr(idx), IDX_I8_M128 if minus_128 else IDX_I8,
                                        x.numel(), _dtype_code(x, 'quantize_to_int8'), C.byref(q), _stream())
        _check(rc, self.lib)
        return idx

    def rowsum_i8(self, w_idx):
        out = torch.empty(w_idx.shape[0], dtype=torch.int32, device=w_idx.device)
        rc = self.lib.tq_rowsum_i8(_ptr(w_idx), _ptr(out), w_idx.shape[0], w_idx.shape[1], _stream())
        _check(rc, self.lib)
        return out

    STAIR_BINS = 768        # fits beside the operand stages of both LDS-tiled kernels (include/tq_hip.h)
    STAIR_BINS_BIG = 1536   # fits the 128 x 128-tile kernel only (grids down to ~0.005 per step instead of ~0.01)

    def stair_bins_for(self, M, N):
        """Bin count for a Linear of M rows and N output features: the launcher's tile rule (csrc/tq_linear_i8.hip,
        launch_linear_t) restated -- a table too large for the kernel that ends up running is ignored there, so a
        mismatch only loses the optimisation."""
        big = M % 128 == 0 and N % 128 == 0 and (M // 128) * (N // 128) >= 1024
        return self.STAIR_BINS_BIG if big else self.STAIR_BINS

    def act_stair(self, activation, q_out, n_bins=None):
        """Staircase table of `activation` followed by the per-tensor <= 8-bit quantizer `q_out` (its 7-tuple) for
        `linear_i8(..., stair=...)`: one launch, no host read; the table's own header says whether it is exact (a grid
        too fine for the bin count makes the consumer keep its arithmetic epilogue).  Valid for the VALUES the range
        buffers hold now: rebuild when they change."""
        n_bins = int(n_bins or self.STAIR_BINS)
        dev = q_out[0].device
        table = torch.empty(self.lib.tq_act_stair_bytes(n_bins), dtype=torch.uint8, device=dev)
        qd = self._qdesc(*q_out, 1, 1)
        rc = self.lib.tq_act_stair_build(int(activation), C.byref(qd), n_bins, table.data_ptr(), table.numel(), _stream())
        _check(rc, self.lib)
        return table, n_bins

    def linear_i8(self, x_idx, w_idx, w_rowsum, bias, x_q, w_delta, w_eps, activation, q_out, out_dtype,
                  want_idx=False, want_y=True, stair=None):
        """x_idx int8 [..., K]; x_q = (delta, zero_float, n_bits, eps) of the input quantizer;
        q_out None or the 7-tuple of a per-tensor quantizer.  -> y [..., N] (, y_idx); want_y=False (needs want_idx):
        index-only output, y is None.  stair: (table, n_bins) of `act_stair(activation, q_out)` or None."""
        K = x_idx.shape[-1]
        M = x_idx.numel() // K
        N = w_idx.shape[0]
        shape = x_idx.shape[:-1] + (N,)
        y = torch.empty(shape, dtype=out_dtype, device=x_idx.device) if want_y else None
        y_idx = torch.empty(shape, dtype=torch.int8, device=x_idx.device) if want_idx else None
        qd = None if q_out is None else self._qdesc(*q_out, 1, 1)
        rc = self.lib.tq_linear_i8_stair_fwd(
            _ptr(x_idx), _ptr(w_idx), _ptr(w_rowsum), _ptr(bias), _ptr(y), _ptr(y_idx), _DTYPES[out_dtype], M, N, K,
            _ptr(x_q[0]), _ptr(x_q[1]), int(x_q[2]), float(x_q[3]), _ptr(w_delta), w_delta.numel(),
            float(w_eps), int(activation), None if qd is None else C.byref(qd),
            None if stair is None else stair[0].data_ptr(), 0 if stair is None else int(stair[1]), _stream())
        _check(rc, self.lib)
        return (y, y_idx) if want_idx else y

    def linear_i8_nonorm(self, x_idx, w_idx, w_rowsum, bias, residual, nn_w, nn_b, x_q, w_delta, w_eps, q_dense, q_sum,
                         q_out, out_dtype, want_idx=False):
        """Integer Linear -> (+ residual) -> NoNorm -> quantizers in one launch; q_* None or 7-tuples. -> y [, y_idx]."""
        K = x_idx.shape[-1]
        M = x_idx.numel() // K
        N = w_idx.shape[0]
        y = torch.empty(x_idx.shape[:-1] + (N,), dtype=out_dtype, device=x_idx.device)
        y_idx = torch.empty(y.shape, dtype=torch.int8, device=y.device) if want_idx else None
        descs = [None if q is None else self._qdesc(*q, 1, 1) for q in (q_dense, q_sum, q_out)]
        refs = [None if d is None else C.byref(d) for d in descs]
        if residual is not None:
            residual = residual.detach().float().contiguous()
        rc = self.lib.tq_linear_i8_nonorm_fwd(
            _ptr(x_idx), _ptr(w_idx), _ptr(w_rowsum), _ptr(bias), _ptr(residual), _ptr(nn_w.detach().float().contiguous()),
            _ptr(nn_b.detach().float().contiguous()), _ptr(y), _ptr(y_idx), _DTYPES[out_dtype], M, N, K, _ptr(x_q[0]),
            _ptr(x_q[1]), int(x_q[2]), float(x_q[3]), _ptr(w_delta), w_delta.numel(), float(w_eps), refs[0], refs[1],
            refs[2], _stream())
        _check(rc, self.lib)
        return (y, y_idx) if want_idx else y

    def linear_i8_nonorm_grouped(self, x_idx, w_idx, w_rowsum, bias, nn_w, nn_b, x_q, w_delta_rows, w_eps, q_dense, q_out,
                                 out_dtype, want_idx=False, n_groups=2):
        """n_groups (2 or 3) Linear -> NoNorm chains reading the same int8 input in one launch (stacked operands, see
        include/tq_hip.h); q_dense / q_out: lists of n_groups 7-tuples or None.  -> [y_g] (, [idx_g]), each [..., N / G]
        contiguous."""
        K = x_idx.shape[-1]
        M = x_idx.numel() // K
        N = w_idx.shape[0]
        G = int(n_groups)
        y = torch.empty((G,) + tuple(x_idx.shape[:-1]) + (N // G,), dtype=out_dtype, device=x_idx.device)
        y_idx = torch.empty(y.shape, dtype=torch.int8, device=y.device) if want_idx else None

        def arr(qs):
            if qs is None:
                return None, None
            descs = [self._qdesc(*q, 1, 1) for q in qs]
            return descs, C.cast((C.POINTER(tq_quantizer) * G)(*[C.pointer(d) for d in descs]), C.POINTER(_QP))
        keep_d, a_d = arr(q_dense)
        keep_o, a_o = arr(q_out)
        rc = self.lib.tq_linear_i8_nonorm_grouped_fwd(
            _ptr(x_idx), _ptr(w_idx), _ptr(w_rowsum), _ptr(bias), _ptr(nn_w), _ptr(nn_b), _ptr(y), _ptr(y_idx),
            _DTYPES[out_dtype], M, N, K, _ptr(x_q[0]), _ptr(x_q[1]), int(x_q[2]), float(x_q[3]), _ptr(w_delta_rows),
            float(w_eps), G, a_d, a_o, _stream())
        _check(rc, self.lib)
        ys = [y[g] for g in range(G)]
        return (ys, [y_idx[g] for g in range(G)]) if want_idx else ys

    def ffn_chain_i8_nonorm(self, x_idx, x_q, residual, stages, out_dtype, want_idx=False):
        """A chain of MobileBERT feed-forward blocks in one launch (include/tq_hip.h tq_ffn_chain_i8_nonorm_fwd).  stages:
        list of dicts with w1_idx, w1_rowsum, bias1, w1_delta, w1_eps, q_mid, w2_idx, w2_rowsum, bias2, w2_delta, w2_eps,
        nn_w, nn_b, q_dense, q_sum, q_out (7-tuples or None).  -> y [, y_idx] of the LAST block."""
        K1 = x_idx.shape[-1]
        M = x_idx.numel() // K1
        N1, N2 = stages[0]['w1_idx'].shape[0], stages[0]['w2_idx'].shape[0]
        y = torch.empty(x_idx.shape[:-1] + (N2,), dtype=out_dtype, device=x_idx.device)
        y_idx = torch.empty(y.shape, dtype=torch.int8, device=y.device) if want_idx else None
        keep = []                                    # descriptors and converted tensors alive across the call
        arr = (tq_ffn_stage * len(stages))()
        for a, g in zip(arr, stages):
            def qp(q):
                if q is None:
                    return None
                d = self._qdesc(*q, 1, 1)
                keep.append(d)
                return C.pointer(d)

            def f32(t):
                t = t.detach().float().contiguous()
                keep.append(t)
                return t.data_ptr()
            a.w1_idx, a.w1_rowsum, a.bias1 = _ptr(g['w1_idx']), _ptr(g['w1_rowsum']), _ptr(g['bias1'])
            a.w1_delta, a.w1_n_params, a.w1_eps = _ptr(g['w1_delta']), g['w1_delta'].numel(), float(g['w1_eps'])
            a.q_mid = qp(g['q_mid'])
            a.w2_idx, a.w2_rowsum, a.bias2 = _ptr(g['w2_idx']), _ptr(g['w2_rowsum']), _ptr(g['bias2'])
            a.w2_delta, a.w2_n_params, a.w2_eps = _ptr(g['w2_delta']), g['w2_delta'].numel(), float(g['w2_eps'])
            a.nn_weight, a.nn_bias = f32(g['nn_w']), f32(g['nn_b'])
            a.q_dense, a.q_sum, a.q_out = qp(g['q_dense']), qp(g['q_sum']), qp(g['q_out'])
        res = residual.detach().float().contiguous()
        rc = self.lib.tq_ffn_chain_i8_nonorm_fwd(_ptr(x_idx), _ptr(x_q[0]), _ptr(x_q[1]), int(x_q[2]), float(x_q[3]), _ptr(res),
                                                 C.cast(arr, C.c_void_p), len(stages), _ptr(y), _ptr(y_idx), _DTYPES[out_dtype],
                                                 M, K1, N1, N2, _stream())
        _check(rc, self.lib)
        return (y, y_idx) if want_idx else y

    FFN_SHAPES = {(128, 512, 128)}          # (K1, N1, N2) tq_ffn_i8_nonorm_fwd is built for

    def ffn_i8_nonorm(self, x_idx, x_q, w1_idx, w1_rowsum, bias1, w1_delta, w1_eps, q_mid, w2_idx, w2_rowsum, bias2, w2_delta,
                      w2_eps, residual, nn_w, nn_b, q_dense, q_sum, q_out, out_dtype, want_idx=False):
        """MobileBERT feed-forward block in one launch (see include/tq_hip.h); q_* 7-tuples or None (q_mid required)."""
        K1 = x_idx.shape[-1]
        M = x_idx.numel() // K1
        N1, N2 = w1_idx.shape[0], w2_idx.shape[0]
        y = torch.empty(x_idx.shape[:-1] + (N2,), dtype=out_dtype, device=x_idx.device)
        y_idx = torch.empty(y.shape, dtype=torch.int8, device=y.device) if want_idx else None
        descs = [None if q is None else self._qdesc(*q, 1, 1) for q in (q_mid, q_dense, q_sum, q_out)]
        refs = [None if d is None else C.byref(d) for d in descs]
        rc = self.lib.tq_ffn_i8_nonorm_fwd(
            _ptr(x_idx), _ptr(x_q[0]), _ptr(x_q[1]), int(x_q[2]), float(x_q[3]), _ptr(w1_idx), _ptr(w1_rowsum), _ptr(bias1),
            _ptr(w1_delta), w1_delta.numel(), float(w1_eps), refs[0], _ptr(w2_idx), _ptr(w2_rowsum), _ptr(bias2),
            _ptr(w2_delta), w2_delta.numel(), float(w2_eps), _ptr(residual.detach().float().contiguous()),
            _ptr(nn_w.detach().float().contiguous()), _ptr(nn_b.detach().float().contiguous()), refs[1], refs[2], refs[3],
            _ptr(y), _ptr(y_idx), _DTYPES[out_dtype], M, K1, N1, N2, _stream())
        _check(rc, self.lib)
        return (y, y_idx) if want_idx else y

    def fake_quant_bwd(self, x, grad_y, delta, zero_float, signed, n_bits, symmetric, log_domain, eps,
                       n_params, inner, param_grads=False):
        _need_device(x, 'fake_quant_bwd')
        x = x.contiguous()
        grad_y = grad_y.contiguous().to(x.dtype)
        if x.dtype == torch.float64:
            d, z = self._as_f64(delta, x), self._as_f64(zero_float, x)
            gx = torch.empty_like(x)
            gd = gz = ws = None
            if param_grads:
                gd = torch.zeros(n_params, dtype=torch.float64, device=x.device)
                gz = torch.zeros(n_params, dtype=torch.float64, device=x.device)
                ws = self._workspace(x.device, self.lib.tq_fake_quant_bwd_f64_workspace_bytes(x.numel(), n_params, inner))
            q = self._qdesc_f64(d, z, signed, n_bits, symmetric, log_domain, eps, n_params, inner)
            rc = self.lib.tq_fake_quant_bwd_f64(_ptr(x), _ptr(grad_y), _ptr(gx), _ptr(gd), _ptr(gz), x.numel(), C.byref(q),
                                                _ptr(ws), ws.numel() if ws is not None else 0, _stream())
            _check(rc, self.lib)
            if gd is not None:        # gradients in the dtype of the buffers they belong to
                gd = gd.to(delta.dtype)
                gz = gz.to(zero_float.dtype) if zero_float is not None else gz
            return gx, gd, gz
        gx = torch.empty_like(x)
        gd = gz = None
        ws = None
        if param_grads:
            gd = torch.zeros(n_params, dtype=torch.float32, device=x.device)
            gz = torch.zeros(n_params, dtype=torch.float32, device=x.device)
            ws = self._workspace(x.device, self.lib.tq_fake_quant_bwd_params_workspace_bytes(x.numel(), n_params, inner))
        q = self._qdesc(delta, zero_float, signed, n_bits, symmetric, log_domain, eps, n_params, inner)
        rc = self.lib.tq_fake_quant_bwd(_ptr(x), _ptr(grad_y), _ptr(gx), _ptr(gd), _ptr(gz), x.numel(),
                                        _dtype_code(x, 'fake_quant_bwd'), C.byref(q), _ptr(ws),
                                        ws.numel() if ws is not None else 0, _stream())
        _check(rc, self.lib)
        return gx, gd, gz

    # -- K4/K5 ---------------------------------------------------------------------------
    def minmax(self, x, n_params=1, inner=1):
        """-> (min, max) fp32; 0-D for n_params == 1 else [n_params]."""
        _need_device(x, 'minmax')
        x = x.detach().contiguous()
        n = x.numel()
        if x.dtype == torch.float64:
            out = torch.empty(2, n_params, dtype=torch.float64, device=x.device)
            ws = self._workspace(x.device, self.lib.tq_minmax_f64_workspace_bytes(n, n_params, inner))
            rc = self.lib.tq_minmax_f64(_ptr(x), n, n_params, inner, _ptr(out[0]), _ptr(out[1]), _ptr(ws), ws.numel(),
                                        _stream())
            _check(rc, self.lib)
            return (out[0, 0], out[1, 0]) if n_params == 1 else (out[0], out[1])
        out = torch.empty(2, n_params, dtype=torch.float32, device=x.device)
        nbytes = self.lib.tq_minmax_workspace_bytes(n, n_params, inner)
        ws = self._workspace(x.device, nbytes)
        rc = self.lib.tq_minmax(_ptr(x), n, _dtype_code(x, 'minmax'), n_params, inner, _ptr(out[0]),
                                _ptr(out[1]), _ptr(ws), ws.numel(), _stream())
        _check(rc, self.lib)
        if n_params == 1:
            return out[0, 0], out[1, 0]
        return out[0], out[1]

    CALIB_MAX_PARAMS = 4096

    def calibrate_minmax(self, x, n_params, inner, mode, prev_min, prev_max, momentum, n_groups, order,
                         n_bits, symmetric, eps, log_domain, want_y=True, out=None):
        """Fused estimating step (statistics -> estimator update -> quantizer parameters -> y).
        -> (cur_min, cur_max, delta, zero_float | None, signed | None, y | None); 0-D for n_params == 1.
        out = (cur_min, cur_max, delta, zero_float | None, signed | None): existing fp32 / bool device
        tensors to update in place (cur_* may be the same tensors as prev_*); fresh tensors otherwise."""
        _need_device(x, 'calibrate_minmax')
        x = x.contiguous()
        dev = x.device
        if n_params == 1 and not n_groups:
            # one range: statistics + update + parameters in one launch (tq_calibrate_tensor)
            st = _stream()
            key = (dev.index, st)
            counter = self._counters.get(key)
            if counter is None:
                counter = self._counters[key] = torch.zeros(1, dtype=torch.int32, device=dev)
            if out is None:
                # cur_min, cur_max, delta, zero_float: one allocation, one op for the four 0-D views
                b0, b1, b2, b3 = torch.empty(4, dtype=torch.float32, device=dev).unbind(0)
                out = (b0, b1, b2, None if symmetric else b3,
                       torch.empty((), dtype=torch.bool, device=dev) if symmetric else None)
            y = torch.empty_like(x) if want_y else None
            ws = self._workspace(dev, self._calib_ws_bytes(x.numel(), 1, 1))
            fc = fastcall()
            if fc is not None and hasattr(fc, 'calibrate_tensor'):
                # the CPython stub: same entry point, no ctypes marshalling of its 21 arguments (~4 us of a ~19 us call)
                addr = self._calib_tensor_addr
                if addr is None:
                    addr = self._calib_tensor_addr = entry_address(self.lib.tq_calibrate_tensor)
                rc = fc.calibrate_tensor(
                    addr, x.data_ptr(), x.numel(), _dtype_code(x, 'calibrate_minmax'), mode, _ptr(prev_min), _ptr(prev_max),
                    out[0].data_ptr(), out[1].data_ptr(), float(momentum), int(n_bits), int(bool(symmetric)), float(eps),
                    int(bool(log_domain)), out[2].data_ptr(), _ptr(out[3]), _ptr(out[4]), _ptr(y), ws.data_ptr(),
                    ws.numel(), counter.data_ptr(), st)
            else:
                rc = self.lib.tq_calibrate_tensor(
                    x.data_ptr(), x.numel(), _dtype_code(x, 'calibrate_minmax'), mode, _ptr(prev_min), _ptr(prev_max),
                    _ptr(out[0]), _ptr(out[1]), float(momentum), int(n_bits), int(bool(symmetric)), float(eps),
                    int(bool(log_domain)), _ptr(out[2]), _ptr(out[3]), _ptr(out[4]), _ptr(y), ws.data_ptr(), ws.numel(),
                    counter.data_ptr(), st)
            if rc != 0:
                _check(rc, self.lib)
            return (*out, y)
        y = torch.empty_like(x) if want_y else None
        if out is not None:
            ws = self._workspace(dev, self._calib_ws_bytes(x.numel(), n_params, inner))
            rc = self.lib.tq_calibrate_minmax(
                _ptr(x), x.numel(), _dtype_code(x, 'calibrate_minmax'), n_params, inner, mode,
                _ptr(prev_min), _ptr(prev_max), _ptr(out[0]), _ptr(out[1]), float(momentum), int(n_groups or 0),
                _ptr(order), int(n_bits), int(bool(symmetric)), float(eps), int(bool(log_domain)),
                _ptr(out[2]), _ptr(out[3]), _ptr(out[4]), _ptr(y), _ptr(ws), ws.numel(), _stream())
            _check(rc, self.lib)
            return (*out, y)
        cur = torch.empty(2, n_params, dtype=torch.float32, device=dev)
        par = torch.empty(1 if symmetric else 2, n_params, dtype=torch.float32, device=dev)
        signed = torch.empty((), dtype=torch.bool, device=dev) if symmetric else None
        ws = self._workspace(dev, self._calib_ws_bytes(x.numel(), n_params, inner))
        rc = self.lib.tq_calibrate_minmax(
            _ptr(x), x.numel(), _dtype_code(x, 'calibrate_minmax'), n_params, inner, mode,
            _ptr(prev_min), _ptr(prev_max), _ptr(cur[0]), _ptr(cur[1]), float(momentum), int(n_groups or 0),
            _ptr(order), int(n_bits), int(bool(symmetric)), float(eps), int(bool(log_domain)),
            _ptr(par[0]), None if symmetric else _ptr(par[1]), _ptr(signed), _ptr(y), _ptr(ws), ws.numel(),
            _stream())
        _check(rc, self.lib)
        if n_params == 1:
            return (cur[0, 0], cur[1, 0], par[0, 0], None if symmetric else par[1, 0], signed, y)
        return (cur[0], cur[1], par[0], None if symmetric else par[1], signed, y)

    def calibrate_stats(self, x, n_params, inner):
        """Sharded calibration, first half: fp32 [2 * n_params] = [-min | max] of the local shard, written by the
        statistics kernel itself into a buffer the caller all-reduces in place (MAX)."""
        _need_device(x, 'calibrate_stats')
        x = x.contiguous()
        dev = x.device
        st = _stream()
        stats = torch.empty(2 * n_params, dtype=torch.float32, device=dev)
        counter = None
        if n_params == 1:
            key = (dev.index, st)
            counter = self._counters.get(key)
            if counter is None:
                counter = self._counters[key] = torch.zeros(1, dtype=torch.int32, device=dev)
        ws = self._workspace(dev, self._calib_ws_bytes(x.numel(), n_params, inner))
        rc = self.lib.tq_calibrate_stats(x.data_ptr(), x.numel(), _dtype_code(x, 'calibrate_stats'), n_params, inner,
                                         stats.data_ptr(), ws.data_ptr(), ws.numel(), _ptr(counter), st)
        _check(rc, self.lib)
        return stats

    def calibrate_apply(self, stats, x, n_params, inner, mode, prev_min, prev_max, momentum, n_groups, order,
                        n_bits, symmetric, eps, log_domain, want_y=True, out=None):
        """Sharded calibration, second half (after the all-reduce of `stats`): estimator update + quantizer
        parameters + y.  Same returns / `out` convention as calibrate_minmax."""
        x = x.contiguous()
        dev = x.device
        if out is None:
            cur = torch.empty(2, n_params, dtype=torch.float32, device=dev)
            par = torch.empty(1 if symmetric else 2, n_params, dtype=torch.float32, device=dev)
            signed = torch.empty((), dtype=torch.bool, device=dev) if symmetric else None
            if n_params == 1:
                out = (cur[0, 0], cur[1, 0], par[0, 0], None if symmetric else par[1, 0], signed)
            else:
                out = (cur[0], cur[1], par[0], None if symmetric else par[1], signed)
        y = torch.empty_like(x) if want_y else None
        rc = self.lib.tq_calibrate_apply(
            stats.data_ptr(), x.data_ptr(), x.numel(), _dtype_code(x, 'calibrate_apply'), n_params, inner, mode,
            _ptr(prev_min), _ptr(prev_max), _ptr(out[0]), _ptr(out[1]), float(momentum), int(n_groups or 0),
            _ptr(order), int(n_bits), int(bool(symmetric)), float(eps), int(bool(log_domain)), _ptr(out[2]),
            _ptr(out[3]), _ptr(out[4]), _ptr(y), _stream())
        _check(rc, self.lib)
        return (*out, y)

    def calibrate_minmax_mailbox(self, box, x, n_params, inner, mode, prev_min, prev_max, momentum, n_groups, order,
                                 n_bits, symmetric, eps, log_domain, want_y=True, out=None):
        """Sharded calibrating step in ONE C call: statistics -> P2P mailbox MAX all-reduce (`box`:
        quantization.mailbox.P2PMailbox) -> estimator update + parameters + y.  Returns like calibrate_minmax."""
        _need_device(x, 'calibrate_minmax_mailbox')
        x = x.contiguous()
        dev = x.device
        st = _stream()
        if out is None:
            cur = torch.empty(2, n_params, dtype=torch.float32, device=dev)
            par = torch.empty(1 if symmetric else 2, n_params, dtype=torch.float32, device=dev)
            signed = torch.empty((), dtype=torch.bool, device=dev) if symmetric else None
            if n_params == 1:
                out = (cur[0, 0], cur[1, 0], par[0, 0], None if symmetric else par[1, 0], signed)
            else:
                out = (cur[0], cur[1], par[0], None if symmetric else par[1], signed)
        counter = None
        if n_params == 1:
            key = (dev.index, st)
            counter = self._counters.get(key)
            if counter is None:
                counter = self._counters[key] = torch.zeros(1, dtype=torch.int32, device=dev)
        y = torch.empty_like(x) if want_y else None
        ws = self._workspace(dev, self._calib_ws_bytes(x.numel(), n_params, inner))
        rc = self.lib.tq_calibrate_minmax_mailbox(
            x.data_ptr(), x.numel(), _dtype_code(x, 'calibrate_minmax_mailbox'), n_params, inner, mode, _ptr(prev_min),
            _ptr(prev_max), _ptr(out[0]), _ptr(out[1]), float(momentum), int(n_groups or 0), _ptr(order), int(n_bits),
            int(bool(symmetric)), float(eps), int(bool(log_domain)), _ptr(out[2]), _ptr(out[3]), _ptr(out[4]), _ptr(y),
            ws.data_ptr(), ws.numel(), _ptr(counter), box.base, box.peers.data_ptr(), box.world, box.rank,
            box.status.data_ptr(), box.spin_budget, st)
        _check(rc, self.lib)
        box.calls += 1
        return (*out, y)

    def calibrate_minmax_rccl(self, comm, x, n_params, inner, mode, prev_min, prev_max, momentum, n_groups, order,
                              n_bits, symmetric, eps, log_domain, want_y=True, out=None):
        """Sharded calibrating step in ONE C call: statistics -> ncclAllReduce(MAX) on the raw communicator `comm`
        (quantization.rccl.RawRcclComm) -> estimator update + parameters + y.  Returns like calibrate_minmax."""
        _need_device(x, 'calibrate_minmax_rccl')
        x = x.contiguous()
        dev = x.device
        st = _stream()
        if out is None:
            signed = torch.empty((), dtype=torch.bool, device=dev) if symmetric else None
            if n_params == 1:
                # cur_min, cur_max, delta, zero_float: one allocation, one op for the four 0-D views
                b0, b1, b2, b3 = torch.empty(4, dtype=torch.float32, device=dev).unbind(0)
            else:
                b0, b1, b2, b3 = torch.empty(4, n_params, dtype=torch.float32, device=dev).unbind(0)
            out = (b0, b1, b2, None if symmetric else b3, signed)
        counter = None
        if n_params == 1:
            key = (dev.index, st)
            counter = self._counters.get(key)
            if counter is None:
                counter = self._counters[key] = torch.zeros(1, dtype=torch.int32, device=dev)
        y = torch.empty_like(x) if want_y else None
        ws = self._workspace(dev, self._calib_ws_bytes(x.numel(), n_params, inner))
        rc = self.lib.tq_calibrate_minmax_rccl(
            x.data_ptr(), x.numel(), _dtype_code(x, 'calibrate_minmax_rccl'), n_params, inner, mode, _ptr(prev_min),
            _ptr(prev_max), _ptr(out[0]), _ptr(out[1]), float(momentum), int(n_groups or 0), _ptr(order), int(n_bits),
            int(bool(symmetric)), float(eps), int(bool(log_domain)), _ptr(out[2]), _ptr(out[3]), _ptr(out[4]), _ptr(y),
            ws.data_ptr(), ws.numel(), _ptr(counter), comm.handle, st)
        _check(rc, self.lib)
        comm.calls += 1
        return (*out, y)

    def range_update(self, mode, new_min, new_max, cur_min, cur_max, momentum=0.9, n_groups=0,
                     order=None):
        """Estimator state update; allocates the state on the first batch. -> (cur_min, cur_max)"""
        _need_device(new_min, 'range_update')
        new_min, new_max = new_min.contiguous(), new_max.contiguous()
        initialised = cur_min is not None
        if not initialised or mode == EST_CURRENT:
            cur_min, cur_max = torch.empty_like(new_min), torch.empty_like(new_max)
        else:
            # the reference rebinds fresh tensors every batch; keep earlier results un-aliased
            cur_min, cur_max = cur_min.clone(), cur_max.clone()
        fn = self.lib.tq_range_update
        if new_min.dtype == torch.float64:
            fn = self.lib.tq_range_update_f64
            new_max, cur_min, cur_max = new_max.double(), cur_min.double(), cur_max.double()
        rc = fn(mode, _ptr(new_min), _ptr(new_max), _ptr(cur_min), _ptr(cur_max), new_min.numel(), int(initialised),
                float(momentum), int(n_groups or 0), _ptr(order), _stream())
        _check(rc, self.lib)
        return cur_min, cur_max

    def axis_ranges(self, new_min, new_max, first):
        _need_device(new_min, 'axis_ranges')
        r = torch.empty_like(new_min)
        fn = self.lib.tq_axis_ranges_f64 if new_min.dtype == torch.float64 else self.lib.tq_axis_ranges
        rc = fn(_ptr(new_min.contiguous()), _ptr(new_max.contiguous()), _ptr(r), r.numel(), int(first), _stream())
        _check(rc, self.lib)
        return r

    def argsort(self, v):
        return torch.argsort(v).contiguous()

    # -- range -> params -------------------------------------------------------------------
    @staticmethod
    def _f64_range(x_min, x_max):
        """float64 range TENSORS keep their dtype (python floats become fp32 tensors, reference quantizers.py:248-250)."""
        return (torch.is_tensor(x_min) and x_min.dtype == torch.float64) or (
            torch.is_tensor(x_max) and x_max.dtype == torch.float64)

    def _set_range_f64(self, x_min, x_max, n_bits, eps, log_domain, symmetric):
        dev = next((t.device for t in (x_min, x_max) if torch.is_tensor(t) and t.is_cuda),
                   torch.device('cuda', torch.cuda.current_device()))
        x_min = torch.as_tensor(x_min).detach().to(device=dev, dtype=torch.float64).contiguous()
        x_max = torch.as_tensor(x_max).detach().to(device=dev, dtype=torch.float64).contiguous()
        delta = torch.empty_like(x_min)
        if symmetric:
            other = torch.empty((), dtype=torch.bool, device=dev)
            fn = self.lib.tq_set_range_sym_f64
        else:
            other = torch.empty_like(x_min)
            fn = self.lib.tq_set_range_asym_f64
        rc = fn(_ptr(x_min), _ptr(x_max), max(x_min.numel(), 1), int(n_bits), float(eps), int(log_domain), _ptr(delta),
                _ptr(other), _stream())
        _check(rc, self.lib)
        return delta, other

    def set_range_asym(self, x_min, x_max, n_bits, eps, log_domain):
        if self._f64_range(x_min, x_max):
            return self._set_range_f64(x_min, x_max, n_bits, eps, log_domain, False)
        x_min = self.to_device_f32(x_min).contiguous()
        x_max = self.to_device_f32(x_max, like=x_min).contiguous()
        delta, zf = torch.empty_like(x_min), torch.empty_like(x_min)
        rc = self.lib.tq_set_range_asym(_ptr(x_min), _ptr(x_max), max(x_min.numel(), 1), int(n_bits),
                                        float(eps), int(log_domain), _ptr(delta), _ptr(zf), _stream())
        _check(rc, self.lib)
        return delta, zf

    def set_range_sym(self, x_min, x_max, n_bits, eps, log_domain):
        if self._f64_range(x_min, x_max):
            return self._set_range_f64(x_min, x_max, n_bits, eps, log_domain, True)
        x_min = self.to_device_f32(x_min).contiguous()
        x_max = self.to_device_f32(x_max, like=x_min).contiguous()
        delta = torch.empty_like(x_min)
        signed = torch.empty((), dtype=torch.bool, device=x_min.device)
        rc = self.lib.tq_set_range_sym(_ptr(x_min), _ptr(x_max), max(x_min.numel(), 1), int(n_bits),
                                       float(eps), int(log_domain), _ptr(delta), _ptr(signed), _stream())
        _check(rc, self.lib)
        return delta, signed

    # -- K7/K8/K9 --------------------------------------------------------------------------
    def mse_candidates(self, x, rows, cand, loss):
        """loss[rows, C] (fp64, device) += sum of squared quantisation error per candidate."""
        _need_device(x, 'mse_candidates')
        x = x.detach().contiguous()
        row_len = x.numel() // rows
        n_cand = cand.shape[0]
        nbytes = self.lib.tq_mse_workspace_bytes(rows, row_len, n_cand)
        ws = self._workspace(x.device, nbytes)
        rc = self.lib.tq_mse_candidates(_ptr(x), rows, row_len, _dtype_code(x, 'mse_candidates'),
                                        _ptr(cand), n_cand, _ptr(loss), _ptr(ws), ws.numel(), _stream())
        _check(rc, self.lib)
        return loss

    def mse_candidates_ordered(self, x, cand, loss=None, per_row=False, want_f32=False):
        """Per-candidate squared quantisation error of x viewed as [len(x), -1], summed in the order of the
        reference's fp32 `torch.sum(torch.sum(err.view(len(data), -1), dim=1))` (range_estimators.py:250-256).
        per_row: keep the row sums (per_channel_loss=True).  loss (fp64 [1 | rows, C], device) gets the fp32
        result ADDED; want_f32 additionally returns the fp32 values themselves.  -> (loss, loss_f32 | None)"""
        _need_device(x, 'mse_candidates_ordered')
        x = x.detach().contiguous()
        rows = x.shape[0] if x.dim() > 0 else 1
        row_len = x.numel() // max(rows, 1)
        n_cand = cand.shape[0]
        out_rows = rows if per_row else 1
        if x.dtype == torch.float64:
            # --double: element arithmetic and sums in float64 (no fp32 value to reproduce); the fp64 cells are the result
            if x.numel():
                rc = self.lib.tq_mse_candidates_f64(_ptr(x), rows, row_len, _ptr(cand), n_cand, int(not per_row), _ptr(loss),
                                                    _stream())
                _check(rc, self.lib)
            return loss, (loss.float() if want_f32 else None)
        f32 = torch.empty((out_rows, n_cand), dtype=torch.float32, device=x.device) if want_f32 else None
        if x.numel() == 0:
            if f32 is not None:
                f32.zero_()
            return loss, f32
        ws = self._workspace(x.device, self.lib.tq_mse_ordered_workspace_bytes(rows, row_len, n_cand))
        rc = self.lib.tq_mse_candidates_ordered(_ptr(x), rows, row_len, _dtype_code(x, 'mse_candidates_ordered'),
                                                _ptr(cand), n_cand, int(not per_row), _ptr(loss), _ptr(f32),
                                                _ptr(ws), ws.numel(), _stream())
        _check(rc, self.lib)
        return loss, f32

    def mse_ordered_plans(self, tensors):
        """Prepared single-candidate launches of `mse_candidates_ordered` for tensors that are evaluated MANY times (the
        lock-step golden-section search: ~20 rounds x 102 weight tensors): shapes, dtype codes, pointers and ONE shared
        workspace (the launches run back to back on one stream) resolved once.  -> (plans, keepalive) or None when a
        tensor needs the generic route (float64, empty, another device)."""
        if not tensors:
            return None
        dev = tensors[0].device
        plans, keep, need = [], [], 0
        for t in tensors:
            if t.dtype == torch.float64 or t.numel() == 0 or t.device != dev or not t.is_cuda:
                return None
            x = t.detach().contiguous()
            rows = x.shape[0] if x.dim() > 0 else 1
            row_len = x.numel() // max(rows, 1)
            need = max(need, int(self.lib.tq_mse_ordered_workspace_bytes(rows, row_len, 1)))
            plans.append((x.data_ptr(), rows, row_len, _dtype_code(x, 'mse_ordered_plans')))
            keep.append(x)
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
        keep.append(ws)
        return [(p + (ws.data_ptr(), ws.numel())) for p in plans], keep

    def mse_ordered_launch(self, plan, cand_ptr, loss_ptr, stream):
        """One candidate (the 4 floats at cand_ptr) for a prepared tensor; its fp32 loss is ADDED to the fp64 cell at loss_ptr."""
        rc = self.lib.tq_mse_candidates_ordered(plan[0], plan[1], plan[2], plan[3], cand_ptr, 1, 1, loss_ptr, None,
                                                plan[4], plan[5], stream)
        if rc != 0:
            _check(rc, self.lib)

    def mse_candidates_grouped(self, x, n_groups, cand, loss):
        """loss[n_groups, C] += per-group (of the LAST axis) squared quantisation error per candidate."""
        _need_device(x, 'mse_candidates_grouped')
        x = x.detach().contiguous()
        d = x.shape[-1]
        n_tokens = x.numel() // d
        n_cand = cand.shape[0]
        nbytes = self.lib.tq_mse_workspace_bytes(n_groups, n_tokens * (d // n_groups), n_cand)
        ws = self._workspace(x.device, nbytes)
        rc = self.lib.tq_mse_candidates_grouped(_ptr(x), n_tokens, d, n_groups,
                                                _dtype_code(x, 'mse_candidates_grouped'), _ptr(cand), n_cand,
                                                _ptr(loss), _ptr(ws), ws.numel(), _stream())
        _check(rc, self.lib)
        return loss

    def xent_candidates(self, x, cand, loss):
        _need_device(x, 'xent_candidates')
        if x.dtype == torch.float64:
            raise TQError('xent_candidates: the cross-entropy range search is fp32 only (no float64 kernel); '
                          'use min-max or MSE estimators with --double')
        x = x.detach().float().contiguous()
        rows = x.shape[0]
        if x.numel() == 0:
            return loss                   # an empty shard adds nothing to the candidate losses
        cols = x.numel() // rows
        rc = self.lib.tq_xent_candidates(_ptr(x), rows, cols, _ptr(cand), cand.shape[0], _ptr(loss),
                                         _stream())
        _check(rc, self.lib)
        return loss

    def argmin_select(self, loss, thr_min, thr_max):
        rows, n_cand = loss.shape
        cur_min = torch.empty(rows, dtype=torch.float32, device=loss.device)
        cur_max = torch.empty(rows, dtype=torch.float32, device=loss.device)
        best = torch.empty(rows, dtype=torch.int64, device=loss.device)
        rc = self.lib.tq_argmin_select(_ptr(loss), rows, n_cand, _ptr(thr_min), _ptr(thr_max),
                                       _ptr(cur_min), _ptr(cur_max), _ptr(best), _stream())
        _check(rc, self.lib)
        return cur_min, cur_max, best

    def candidate_table(self, table_np, device):
        """host numpy [C,k] fp32 -> device tensor (one async H2D copy)."""
        return torch.from_numpy(table_np).to(device, non_blocking=True)

    def zeros_f64(self, shape, device):
        return torch.zeros(shape, dtype=torch.float64, device=device)

    # -- K10/K11/K13 -----------------------------------------------------------------------
    def adaround_fwd(self, w, alpha, qargs, mode, soft, temperature):
        _need_device(w, 'adaround_fwd')
        _need_f32('adaround_fwd', w, alpha)
        w = w.detach().contiguous()
        out = torch.empty_like(w)
        q = self._qdesc(*qargs)
        rc = self.lib.tq_adaround_fwd(_ptr(w), _ptr(alpha), _ptr(out), w.numel(), C.byref(q), mode,
                                      int(soft), float(temperature or 0.0), _stream())
        _check(rc, self.lib)
        return out

    def adaround_init_alpha(self, w, qargs, mode, temperature):
        _need_device(w, 'adaround_init_alpha')
        _need_f32('adaround_init_alpha', w)
        w = w.detach().contiguous()
        alpha = torch.empty_like(w)
        q = self._qdesc(*qargs)
        rc = self.lib.tq_adaround_init_alpha(_ptr(w), _ptr(alpha), w.numel(), C.byref(q), mode,
                                             float(temperature or 0.0), _stream())
        _check(rc, self.lib)
        return alpha

    def adaround_bwd(self, w, alpha, grad_wq, qargs, mode, temperature):
        _need_device(w, 'adaround_bwd')
        _need_f32('adaround_bwd', w, alpha)
        grad_wq = grad_wq.float()
        g = torch.empty_like(alpha)
        q = self._qdesc(*qargs)
        rc = self.lib.tq_adaround_bwd(_ptr(w.detach().contiguous()), _ptr(alpha.detach().contiguous()),
                                      _ptr(grad_wq.contiguous()), _ptr(g), alpha.numel(), C.byref(q),
                                      mode, float(temperature or 0.0), _stream())
        _check(rc, self.lib)
        return g

    def adaround_bwd_adam(self, w, grad_wq, alpha, exp_avg, exp_avg_sq, qargs, mode, temperature,
                          reg_weight, beta, lr, b1, b2, adam_eps, step, want_grad=False):
        _need_device(w, 'adaround_bwd_adam')
        _need_f32('adaround_bwd_adam', w, alpha, exp_avg, exp_avg_sq)
        grad_wq = grad_wq.float()
        g = torch.empty_like(alpha) if want_grad else None
        q = self._qdesc(*qargs)
        rc = self.lib.tq_adaround_bwd_adam(_ptr(w.detach().contiguous()), _ptr(grad_wq.contiguous()),
                                           _ptr(alpha), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(g),
                                           alpha.numel(), C.byref(q), mode, float(temperature or 0.0),
                                           float(reg_weight), float(beta), float(lr), float(b1),
                                           float(b2), float(adam_eps), int(step), _stream())
        _check(rc, self.lib)
        return g

    def adaround_bwd_adam_sched(self, w, grad_wq, alpha, exp_avg, exp_avg_sq, qargs, mode, temperature, sched, lr, b1, b2,
                                adam_eps):
        """K11 with (reg_weight, beta, 1 - b1^t, sqrt(1 - b2^t)) read from the fp32 device tensor `sched`."""
        _need_device(w, 'adaround_bwd_adam_sched')
        _need_f32('adaround_bwd_adam_sched', w, alpha, exp_avg, exp_avg_sq, sched, grad_wq)
        q = self._qdesc(*qargs)
        rc = self.lib.tq_adaround_bwd_adam_sched(_ptr(w.detach().contiguous()), _ptr(grad_wq.contiguous()), _ptr(alpha),
                                                 _ptr(exp_avg), _ptr(exp_avg_sq), alpha.numel(), C.byref(q), mode,
                                                 float(temperature or 0.0), _ptr(sched.contiguous()), float(lr), float(b1),
                                                 float(b2), float(adam_eps), _stream())
        _check(rc, self.lib)

    def order_stats(self, rows2d, ranks):
        """[rows, len(ranks)] fp32: the elements of the given 0-based ascending ranks of every row (NaN last), selected
        on the device without sorting (tq_order_stats: radix select; the percentile estimators need two order
        statistics per percentile, reference range_estimators.py:121-140)."""
        _need_device(rows2d, 'order_stats')
        x = rows2d.detach().contiguous()
        rows, n = x.shape
        m = len(ranks)
        out = torch.empty(rows, m, dtype=torch.float32, device=x.device)
        ws = self._workspace(x.device, self.lib.tq_order_stats_workspace_bytes(rows, m))
        rk = (C.c_uint64 * m)(*[int(r) for r in ranks])
        rc = self.lib.tq_order_stats(_ptr(x), rows, n, _dtype_code(x, 'order_stats'), rk, m, _ptr(out), _ptr(ws),
                                     ws.numel() * ws.element_size(), _stream())
        _check(rc, self.lib)
        return out

    def adaround_reg(self, alpha, mode, temperature, beta, weight):
        _need_device(alpha, 'adaround_reg')
        _need_f32('adaround_reg', alpha)
        out = torch.zeros(1, dtype=torch.float64, device=alpha.device)
        ws = self._workspace(alpha.device, self.lib.tq_reduce_workspace_bytes(alpha.numel()))
        rc = self.lib.tq_adaround_reg(_ptr(alpha.detach().contiguous()), alpha.numel(), mode,
                                      float(temperature or 0.0), float(beta), float(weight), _ptr(out),
                                      _ptr(ws), ws.numel(), _stream())
        _check(rc, self.lib)
        return out[0]

    def recon_loss(self, pred, tgt):
        _need_device(pred, 'recon_loss')
        pred, tgt = pred.detach().float().contiguous(), tgt.detach().float().contiguous()
        d0 = pred.shape[0]
        d1 = pred.shape[1] if pred.dim() > 1 else 1
        rest = pred.numel() // (d0 * d1)
        out = torch.empty(1, dtype=torch.float64, device=pred.device)
        ws = self._workspace(pred.device, self.lib.tq_reduce_workspace_bytes(pred.numel()))
        rc = self.lib.tq_recon_loss(_ptr(pred), _ptr(tgt), d0, d1, rest, _ptr(out), _ptr(ws), ws.numel(),
                                    _stream())
        _check(rc, self.lib)
        return out[0]


for _name, _fn in list(vars(HipBackend).items()):
    if callable(_fn) and not _name.startswith('_') and _name not in ('candidate_table', 'zeros_f64', 'argsort'):
        setattr(HipBackend, _name, _device_guarded(_fn))
del _name, _fn

_backend = None


def backend():
    """The active backend (HipBackend unless a test installed a double with set_backend)."""
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend


def raise_deferred(sync=False):
    """Raise what queued launches could only report asynchronously (HipBackend.raise_deferred); a no-op before the first
    launch and for backend doubles."""
    f = getattr(_backend, 'raise_deferred', None)
    if f is not None:
        f(sync)


def on_device(t):
    """True when the active backend can take `t` where it lives: ROCm tensors for the HIP backend (there is no CPU
    fallback); a test double that declares `accepts_cpu` (tests/_oracle_backend.py) also takes host tensors, so that the
    host logic of the fused / integer paths can be replayed against the CPU oracle."""
    return t.is_cuda or getattr(backend(), 'accepts_cpu', False)


def set_backend(b):
    """Test hook: install a backend double (or None to go back to HIP)."""
    global _backend
    prev, _backend = _backend, b
    return prev
