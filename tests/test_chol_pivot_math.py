"""The pivot arithmetic of the Cholesky kernel (vbmc_amd/csrc/chol_mfma.h: chol_sqrt_rsqrt) restated with exactly rounded
fused multiply-adds: from a reciprocal-root seed that is only good to 2^-20 (v_rsq_f64 is better), two Goldschmidt steps and the
residual correction give sqrt(p) to 1 ulp and 1/sqrt(p) to 2 ulp -- the precision the factorisation's error analysis assumes.
CPU only; the kernel itself is checked against a long-double Cholesky by tools/chol_bench.hip and through the GPU tests."""
import math
from fractions import Fraction

import numpy as np


def fma(a, b, c):
    """round(a * b + c) with a single rounding (exact rational arithmetic, then round-to-nearest-even)"""
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def chol_sqrt_rsqrt(p, seed_rel_err):
    y = (1.0 / math.sqrt(p)) * (1.0 + seed_rel_err)      # stand-in for v_rsq_f64
    g, h = p * y, 0.5 * y
    r = fma(-h, g, 0.5)
    g, h = fma(g, r, g), fma(h, r, h)
    r = fma(-h, g, 0.5)
    g, h = fma(g, r, g), fma(h, r, h)
    d = fma(-g, g, p)
    return fma(d, h, g), h + h


def ulps(x, exact):
    return abs(Fraction(x) - exact) / Fraction(math.ulp(float(exact)))


def test_goldschmidt_pivot_root_and_reciprocal_root():
    rng = np.random.default_rng(7)
    worst_rs = worst_ri = Fraction(0)
    for _ in range(1500):
        p = float(np.exp(rng.uniform(np.log(1e-12), np.log(1e12))))
        e = float(rng.uniform(-1.0, 1.0)) * 2.0 ** -20
        rs, ri = chol_sqrt_rsqrt(p, e)
        # exact references to ~1e-30: Newton on rationals from the double root
        s = Fraction(math.sqrt(p))
        for _ in range(3):
            s = (s + Fraction(p) / s) / 2
        worst_rs = max(worst_rs, ulps(rs, s))
        worst_ri = max(worst_ri, ulps(ri, 1 / s))
    assert worst_rs <= 1 and worst_ri <= 2, (float(worst_rs), float(worst_ri))


def test_pivot_times_reciprocal_root_reproduces_the_root():
    """the column scaling uses w * ri where the diagonal gets rs: the two must agree to rounding (p * ri vs rs)"""
    rng = np.random.default_rng(8)
    for _ in range(300):
        p = float(np.exp(rng.uniform(np.log(1e-8), np.log(1e8))))
        rs, ri = chol_sqrt_rsqrt(p, 2.0 ** -21)
        assert abs(p * ri - rs) <= 4 * math.ulp(rs)
