"""A/B of two builds of libtq_hip.so on the SAME box: per-tensor bf16 fake-quant of the bench tensor."""
import ctypes as C, sys, time
sys.path.insert(0, '/root/repo/transformer-quantization_amd')
import torch
from quantization import _hip
old = C.CDLL('/root/repo/transformer-quantization_amd/lib/libtq_old.so')
old.tq_fake_quant_fwd.restype = C.c_int
old.tq_fake_quant_fwd.argtypes = _hip.SIGNATURES['tq_fake_quant_fwd'][1]
libs = {'new': _hip.load_library(), 'old': old}
x = (torch.randn(1024, 512, 768, device='cuda') * 3).to(torch.bfloat16)
y = torch.empty_like(x)
delta = torch.tensor(0.05, device='cuda'); zf = torch.tensor(120.0, device='cuda')
q = _hip.tq_quantizer(delta.data_ptr(), zf.data_ptr(), None, 8, 0, 0, 1e-8, 1, 1)
st = torch.cuda.current_stream().cuda_stream
def run(lib, n):
    for _ in range(n):
        lib.tq_fake_quant_fwd(x.data_ptr(), y.data_ptr(), None, 0, x.numel(), 1, C.byref(q), st)
t0 = time.time()
while time.time() - t0 < 0.5: run(libs['new'], 10)
def timeit(f):
    f(5)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record(); f(50); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 50 * 1e3
for rnd in range(3):
    print(rnd, 'per-tensor bf16, indices span the whole grid:', ' '.join(f'{n} {timeit(lambda k: run(libs[n], k)):.1f} us' for n in ('old', 'new')))
# per-embedding bf16 (d = 768), every column's indices span the whole grid
d = 768
dl = (torch.rand(d, device='cuda') * 0.02 + 0.04); zfs = torch.rand(d, device='cuda') * 100 + 70
qa = _hip.tq_quantizer(dl.data_ptr(), zfs.data_ptr(), None, 8, 0, 0, 1e-8, d, 1)
def runa(lib, n):
    for _ in range(n):
        lib.tq_fake_quant_fwd(x.data_ptr(), y.data_ptr(), None, 0, x.numel(), 1, C.byref(qa), st)
for rnd in range(3):
    print(rnd, 'per-embedding bf16, full-span indices:', ' '.join(f'{n} {timeit(lambda k: runa(libs[n], k)):.1f} us' for n in ('old', 'new')))
# bench-like data: two outlier dims stretch the range, most indices small
xb = torch.randn(1024, 512, 768, device='cuda'); xb[..., 308] *= 20; xb[..., 381] *= 20; xb = xb.to(torch.bfloat16)
x.copy_(xb)
for rnd in range(2):
    print(rnd, 'per-embedding bf16, bench-like data:', ' '.join(f'{n} {timeit(lambda k: runa(libs[n], k)):.1f} us' for n in ('old', 'new')))
