"""The C-ABI library loads on a CPU-only box and exports exactly what include/tq_hip.h declares
(no compute calls here)."""
import ctypes
import os
import re
import subprocess

import pytest

from tests.conftest import ROOT, PKG

HEADER = os.path.join(ROOT, 'include', 'tq_hip.h')
LIB = os.path.join(PKG, 'lib', 'libtq_hip.so')


def _declared():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(tq_[a-z0-9_]+)\s*\(', src)))


RESOURCES = os.path.join(PKG, 'lib', 'kernel_resources.json')


@pytest.fixture(scope='module')
def lib_path():
    if not os.path.exists(LIB) or not os.path.exists(RESOURCES):
        import importlib.util
        spec = importlib.util.spec_from_file_location('tq_build', os.path.join(PKG, 'build.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build(verbose=False)
    return LIB


def test_header_symbols_are_exported(lib_path):
    declared = _declared()
    assert len(declared) >= 20
    lib = ctypes.CDLL(lib_path)
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in tq_hip.h but not exported'
    out = subprocess.check_output(['nm', '-D', '--defined-only', lib_path], text=True)
    exported = sorted(set(re.findall(r' T (tq_[a-z0-9_]+)', out)))
    assert exported == declared, (set(exported) ^ set(declared))


def test_python_binding_covers_header(lib_path):
    from quantization import _hip
    assert sorted(_hip.SIGNATURES) == _declared()
    lib = _hip.load_library()
    header = open(os.path.join(ROOT, 'include', 'tq_hip.h')).read()
    assert lib.tq_abi_version() == _hip.ABI_VERSION == int(re.search(r'#define TQ_ABI_VERSION (\d+)', header).group(1))
    assert lib.tq_last_error() is not None


def test_gfx950_code_object_present(lib_path):
    blob = open(lib_path, 'rb').read()
    assert b'gfx950' in blob


def test_missing_library_fails_loudly(tmp_path):
    from quantization import _hip
    with pytest.raises(_hip.TQError):
        _hip.load_library(str(tmp_path / 'nope.so'))


def test_argument_validation_without_gpu(lib_path):
    """Validation happens before any HIP call, so it is testable without a device."""
    from quantization import _hip
    lib = _hip.load_library()
    q = _hip.tq_quantizer(None, None, None, 8, 0, 0, 1e-8, 1, 1)
    rc = lib.tq_fake_quant_fwd(None, None, None, 0, 16, 0, ctypes.byref(q), None)
    assert rc == -1 and b'NULL' in lib.tq_last_error()
    assert lib.tq_fake_quant_fwd(None, None, None, 0, 0, 0, ctypes.byref(q), None) == 0   # empty
    assert lib.tq_minmax_workspace_bytes(1 << 20, 1, 1) > 0
    assert lib.tq_minmax_workspace_bytes(1024 * 768, 768, 1) >= 2 * 768 * 4
    assert lib.tq_mse_workspace_bytes(1, 786432, 100) >= 100 * 8


def test_quantizer_descriptors_are_memoised_by_value(lib_path):
    """The C descriptor of a quantizer is memoised on (pointers, scalars): the same operands give the same struct, a
    rebound range tensor (what every calibration step does) or any changed scalar a different one with the new fields."""
    import torch
    from quantization import _hip
    be = _hip.HipBackend()
    d1, z1, d2 = torch.tensor(0.1), torch.tensor(3.0), torch.tensor(0.2)
    a = be._qdesc(d1, z1, None, 8, False, False, 1e-8, 1, 1)
    assert be._qdesc(d1, z1, None, 8, False, False, 1e-8, 1, 1) is a
    assert a.delta == d1.data_ptr() and a.zero_float == z1.data_ptr() and not a.signed_flag
    assert (a.n_bits, a.symmetric, a.log_domain, a.n_params, a.inner) == (8, 0, 0, 1, 1) and abs(a.eps - 1e-8) < 1e-15
    b = be._qdesc(d2, z1, None, 8, False, False, 1e-8, 1, 1)
    assert b is not a and b.delta == d2.data_ptr()
    for changed in ((d1, z1, None, 4, False, False, 1e-8, 1, 1), (d1, z1, None, 8, True, False, 1e-8, 1, 1),
                    (d1, z1, None, 8, False, True, 1e-8, 1, 1), (d1, z1, None, 8, False, False, 1e-6, 1, 1),
                    (d1, z1, None, 8, False, False, 1e-8, 768, 1), (d1, z1, None, 8, False, False, 1e-8, 768, 64),
                    (d1, None, d2, 8, True, False, 1e-8, 1, 1)):
        c = be._qdesc(*changed)
        assert c is not a
        assert (c.n_bits, c.symmetric, c.log_domain, c.n_params, c.inner) == (
            changed[3], int(changed[4]), int(changed[5]), changed[7], changed[8])
    assert be._calib_ws_bytes(1 << 20, 1, 1) == be.lib.tq_calibrate_workspace_bytes(1 << 20, 1, 1)


def test_product_package_never_imports_oracle():
    bad = []
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith('.py'):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M) or 'tq_oracle' in txt:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_fastcall_stub_calls_the_same_library_through_raw_addresses():
    """csrc_py/tq_fastcall.c (plain C over Python.h: no torch headers, no pybind) is built by build.py next to
    libtq_hip.so; it reaches the SAME entry points through their raw addresses.  No compute call here (no GPU): the
    version getter and an argument-validation error path are enough to prove the route."""
    from quantization import _hip
    fc = _hip.fastcall()
    assert fc is not None, 'lib/_tq_fastcall*.so missing: python transformer-quantization_amd/build.py'
    lib = _hip.load_library()
    assert fc.call_ptrs(_hip.entry_address(lib.tq_abi_version)) == lib.tq_abi_version() == _hip.ABI_VERSION
    # tq_fake_quant_fwd(x = NULL, ...) is rejected by the library's own argument checks on both routes
    addr = _hip.entry_address(lib.tq_fake_quant_fwd)
    rc_fast = fc.fake_quant_fwd(addr, 0, 0, 0, 0, 16, 0, 0, 0)
    msg_fast = lib.tq_last_error()
    rc_ctypes = lib.tq_fake_quant_fwd(None, None, None, 0, 16, 0, None, None)
    assert rc_fast == rc_ctypes != 0 and msg_fast == lib.tq_last_error()
    import pytest
    with pytest.raises(ValueError):
        fc.fake_quant_fwd(0, 0, 0, 0, 0, 16, 0, 0, 0)
    with pytest.raises(TypeError):
        fc.fake_quant_fwd(addr, 0)
    # the calibrating call: 21 arguments, two of them floating point -- the NULL-pointer rejection comes back through both
    # routes with the same message
    caddr = _hip.entry_address(lib.tq_calibrate_tensor)
    rc_fast = fc.calibrate_tensor(caddr, 0, 16, 0, 0, None, None, 0, 0, 0.1, 8, 0, 1e-8, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    msg_fast = lib.tq_last_error()
    rc_ctypes = lib.tq_calibrate_tensor(None, 16, 0, 0, None, None, None, None, 0.1, 8, 0, 1e-8, 0, None, None, None, None, None, 0,
                                        None, None)
    assert rc_fast == rc_ctypes != 0 and msg_fast == lib.tq_last_error()
    with pytest.raises(TypeError):
        fc.calibrate_tensor(caddr, 0, 16)
    with pytest.raises(TypeError):
        fc.calibrate_tensor(caddr, 0, 16, 0, 0, None, None, 0, 0, 'momentum', 8, 0, 1e-8, 0, 0, 0, 0, 0, 0, 0, 0, 0)


@pytest.mark.gpu
def test_fixed_range_fast_path_is_identical_through_fastcall_and_ctypes():
    import torch
    from quantization import _hip
    from quantization.base_quantized_classes import QuantizedActivation
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    x = torch.randn(8, 128, 768, device='cuda')
    outs, cal = {}, {}
    saved = (_hip._fastcall_mod, _hip._fastcall_tried)
    try:
        for route in ('fastcall', 'ctypes'):
            if route == 'ctypes':
                _hip._fastcall_mod, _hip._fastcall_tried = None, True
            qa = QuantizedActivation(act_method=QMethods.asymmetric_uniform, n_bits_act=8,
                                     act_range_method=RangeEstimators.running_minmax).cuda()
            qa.quantized_acts()
            qa.eval()
            with torch.no_grad():
                for _ in range(3):                      # calibrating calls: tq_calibrate_tensor through the same route
                    y_cal = qa(x * 1.5) if _ == 1 else qa(x)
                est = qa.activation_quantizer.range_estimator
                cal[route] = (y_cal, est.current_xmin.clone(), est.current_xmax.clone(),
                              qa.activation_quantizer.quantizer._delta.clone())
                qa.activation_quantizer.fix_ranges()
                outs[route] = qa(x)
                plan = qa.activation_quantizer._fast_plan
            assert plan is not None and plan[9] is not None and (type(plan[10]) is tuple) == (route == 'fastcall')
    finally:
        _hip._fastcall_mod, _hip._fastcall_tried = saved
    assert torch.equal(outs['fastcall'], outs['ctypes'])
    assert all(torch.equal(a, b) for a, b in zip(cal['fastcall'], cal['ctypes']))


def test_no_kernel_uses_scratch_memory(lib_path):
    """build.py records what the compiler reports for EVERY kernel of the library (lib/kernel_resources.json).  None of
    them may spill: in round 6 a change of a shared header made the register allocator put the accumulators of the 128 x 128
    tile integer Linear into scratch memory (272 bytes per lane, M = 8192: 45 -> 114 us) and only the kernel table showed
    it, two collections later.  Also: the launch bounds leave no kernel without a resident wave."""
    import json
    with open(RESOURCES) as f:
        res = json.load(f)
    assert len(res) >= 400, len(res)
    spilling = {k: v['scratch'] for k, v in res.items() if v.get('scratch', 0) > 0}
    assert not spilling, spilling
    assert all(v.get('occupancy', 1) >= 1 and v.get('vgpr', 0) + v.get('agpr', 0) <= 512 for v in res.values())
