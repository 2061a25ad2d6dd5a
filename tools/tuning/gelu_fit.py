#!/usr/bin/env python3
"""The single-fit erf behind the integer GEMM's GELU epilogue (csrc/tq_linear_i8.hip `gelu_erf_n`, oracle/tq_int_oracle.c
`gelu_fit`): erfc(t) = 2^(-t Q(t)) on t = |v| / sqrt 2 in [0, 4], Q of degree 7, fitted for a uniform ABSOLUTE error of erf
(iteratively re-weighted least squares towards the minimax solution; weight = d erf / d exponent = erfc ln 2).

Prints the coefficients, the error of the fp32 evaluation against scipy's erf, the error of the resulting GELU against
float64 next to the error of the reference's own fp32 evaluation (correctly rounded erf, same operation order as
nn.GELU()), and the fraction of 8-bit output indices that differ between the two.  CPU only (numpy + scipy)."""
import numpy as np
from scipy.special import erf, erfc

T, DEG = 4.0, 8                      # P(t) = t (c0 + ... + c7 t^7)
t = np.concatenate([np.linspace(1e-7, 0.5, 60001), np.linspace(0.5, T, 140001)])
g = -np.log2(erfc(t))
w = erfc(t) * np.log(2)
A = np.vstack([t ** (k + 1) for k in range(DEG)]).T
ww, best = w.copy(), None
for _ in range(400):
    c, *_ = np.linalg.lstsq(A * ww[:, None], g * ww, rcond=None)
    err = (A @ c - g) * w
    m = np.abs(err).max()
    if best is None or m < best[0]:
        best = (m, c.copy())
    ww = ww * (1 + 0.3 * np.abs(err) / m)
fit_err, c = best
c32 = c.astype(np.float32)
print('fit error of erf (float64 evaluation):', fit_err)
print('Q coefficients c0..c7 (fp32):', [float(x) for x in c32])
assert c32[-1] > 0, 'the exponent must keep growing beyond the fitted interval'


def erfc32(tt):
    tt = tt.astype(np.float32)
    q = np.full_like(tt, -c32[-1])
    for k in range(DEG - 2, -1, -1):
        q = (q.astype(np.float64) * tt - np.float64(c32[k])).astype(np.float32)      # fma
    p = (q * tt).astype(np.float32)
    return np.exp2(p.astype(np.float64)).astype(np.float32)


print('erfc, fp32 evaluation: max abs error', np.abs(erfc32(t).astype(np.float64) - erfc(t)).max())
rs = np.random.RandomState(0)
v = (rs.standard_normal(4_000_000) * 1.5).astype(np.float32)
a = (v * np.float32(0.70710678118654752440)).astype(np.float32)
e = (np.float32(1) - erfc32(np.abs(a))).astype(np.float32)
r = np.copysign(e, a)
gel = ((v * np.float32(0.5)).astype(np.float32) * (np.float32(1) + r).astype(np.float32)).astype(np.float32)
truth = 0.5 * v.astype(np.float64) * (1 + erf(v.astype(np.float64) / np.sqrt(2)))
ref = ((v * np.float32(0.5)).astype(np.float32) * (np.float32(1) + erf(a.astype(np.float64)).astype(np.float32)).astype(np.float32)).astype(np.float32)
print('GELU max abs error vs float64: this fit', np.abs(gel - truth).max(), '| fp32 evaluation with a correctly rounded erf',
      np.abs(ref - truth).max())
for scale in (0.01, 0.024, 0.05):
    i1, i2 = np.rint(gel / np.float32(scale)), np.rint(ref / np.float32(scale))
    print(f'8-bit grid step {scale}: indices that differ {np.mean(i1 != i2):.2e}, max distance {int(np.abs(i1 - i2).max())}')
