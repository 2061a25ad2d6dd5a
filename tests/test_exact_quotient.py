"""The two identities behind the branch-free quantizer of the gfx950 kernels (csrc/tq_device.h, `QF`), checked with
EXACT rational arithmetic on adversarial inputs (rounding ties +- 2 ulp, grid ends, scales whose significand is all ones):

 (1) r = RN(1/s), q0 = RN(x r), e = RN(x - q0 s) [fma], q1 = RN(q0 + e r) [fma]   ==>  q1 == RN(x / s)
 (2) Q(x) = rne(RN(x/s)):  Q(med3(x, RN(s k_lo), RN(s k_hi))) == clamp(Q(x), k_lo, k_hi)

CPU only, no kernel involved: this pins the arithmetic the kernels rely on; the kernels themselves are compared with the
oracle's true-division chain on tie-adjacent inputs by tests/test_hip_parity.py::test_rounding_ties_are_bit_exact."""
import random
from fractions import Fraction

import numpy as np

F32 = np.float32


def rn(fr):
    """Fraction -> nearest-even float32 (normal range)."""
    if fr == 0:
        return F32(0.0)
    sign = -1 if fr < 0 else 1
    a = abs(fr)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    if Fraction(2) ** (e + 1) <= a:
        e += 1
    e = max(e, -126)
    ulp = Fraction(2) ** (e - 23)
    n = a / ulp
    fl = n.numerator // n.denominator
    rem = n - fl
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and fl % 2 == 1):
        fl += 1
    return F32(sign * float(fl * ulp))


def fr(x):
    return Fraction(float(x))


def fma(a, b, c):
    return rn(fr(a) * fr(b) + fr(c))


def rand_scale(rng):
    c = rng.random()
    if c < 0.2:
        m = (1 << 24) - rng.randint(1, 4)          # significand all ones (the classic hard case for reciprocals)
    elif c < 0.4:
        m = (1 << 23) + rng.randint(0, 4)
    else:
        m = rng.randint(1 << 23, (1 << 24) - 1)
    return F32(float(Fraction(m) * Fraction(2) ** (rng.randint(-40, 10) - 23)))


def ulp_step(x, n):
    for _ in range(abs(n)):
        x = np.nextafter(x, F32(np.inf if n > 0 else -np.inf))
    return x


def test_fma_refined_quotient_is_correctly_rounded():
    rng = random.Random(1)
    n = 0
    for _ in range(4000):
        s = rand_scale(rng)
        k = rng.choice([rng.randint(-300, 300), rng.randint(-70000, 70000), rng.randint(-(1 << 22), 1 << 22)])
        x0 = rn((Fraction(k) + Fraction(1, 2)) * fr(s))                 # next to a rounding tie of x / s
        for du in (-2, -1, 0, 1, 2):
            x = ulp_step(x0, du)
            r = rn(Fraction(1) / fr(s))
            q0 = rn(fr(x) * fr(r))
            q1 = fma(fma(-q0, s, x), r, q0)
            assert q1 == rn(fr(x) / fr(s)), (float(x), float(s))
            n += 1
    assert n == 20000


def test_clamping_the_operand_equals_clamping_the_index():
    rng = random.Random(2)

    def Q(x, s):
        return np.rint(rn(fr(x) / fr(s)))
    for _ in range(1500):
        s = rand_scale(rng)
        nb = rng.choice([2, 4, 8, 8, 16, 20])
        if rng.random() < 0.5:
            lo, hi = 0, 2 ** nb - 1
            zp = rng.randint(lo, hi)
        else:
            lo, hi, zp = -2 ** (nb - 1), 2 ** (nb - 1) - 1, 0
        klo, khi = lo - zp, hi - zp
        ylo, yhi = F32(s * F32(klo)), F32(s * F32(khi))
        for k in (klo, khi, rng.randint(klo, khi)):
            for off in (-1.5, -0.5, 0.0, 0.5, 1.5):
                x0 = rn((Fraction(k) + Fraction(off)) * fr(s))
                for du in (-1, 0, 1):
                    x = ulp_step(x0, du)
                    ref = min(max(Q(x, s), klo), khi)
                    assert Q(min(max(x, ylo), yhi), s) == ref, (float(x), float(s), klo, khi)


def test_dequantisation_as_one_fma_normalises_the_zero():
    """Round 4: y = fma(s, h, +0) replaces y = s * (h + 0) (tq_device.h `qf_dequant2`).  For h != 0 adding an exact zero to
    the exact product changes nothing, so RN(s h + 0) == RN(s h) == the fp32 product; for h = -0 the exact product is
    -0 and (-0) + (+0) = +0 under round-to-nearest -- the +0 the reference's `scale * (x_int - zero_point)` yields.
    float64 holds the product of two fp32 numbers exactly, so `np.float32(float64 product + 0.0)` IS the fused result."""
    import numpy as np
    rs = np.random.RandomState(3)
    s = np.concatenate([rs.uniform(1e-6, 4.0, 20000), 2.0 ** rs.uniform(-90, 90, 20000)]).astype(np.float32)
    h = np.concatenate([rs.randint(-(1 << 21), 1 << 21, 20000), rs.randint(-255, 256, 20000)]).astype(np.float32)
    fused = (s.astype(np.float64) * h.astype(np.float64) + np.float64(0.0)).astype(np.float32)
    two_step = s * (h + np.float32(0.0))
    assert np.array_equal(fused.view(np.uint32), two_step.view(np.uint32))
    neg_zero = np.float32(-0.0)
    z = (s.astype(np.float64) * np.float64(neg_zero) + np.float64(0.0)).astype(np.float32)
    assert not np.signbit(z).any() and np.array_equal(z, np.zeros_like(z))
    assert np.signbit(s * neg_zero).all()                 # what the bare product would have returned
