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


def _blocked_chol_model(A, refine, two_panel):
    """NumPy model of k_chol2's algorithm (16-row block steps, panel = inv(Rkk') A12 as a PRODUCT with the explicit inverse,
    optionally one step of iterative refinement, trailing updates deferred over two panels when two_panel) -- the arithmetic
    order differs from the kernel's, the error mechanism is the same."""
    A = np.array(A, dtype=np.float64)
    N = A.shape[0]
    R = np.zeros_like(A)
    pend = []                                   # panels not yet applied to the trailing matrix beyond the next row block
    for k0 in range(0, N, 16):
        k1 = min(k0 + 16, N)
        Rkk = np.linalg.cholesky(A[k0:k1, k0:k1]).T
        R[k0:k1, k0:k1] = Rkk
        if k1 == N:
            break
        W = np.linalg.inv(Rkk.T)                # explicit inverse of the diagonal factor
        P = W @ A[k0:k1, k1:]
        if refine:
            P = P + W @ (A[k0:k1, k1:] - Rkk.T @ P)
        R[k0:k1, k1:] = P
        pend.append((k1, P))
        n1 = min(k1 + 16, N)
        # the next row block (and its diagonal tile) always sees every pending panel before it is factored
        for (c0, Pp) in pend:
            off = k1 - c0
            A[k1:n1, k1:] -= Pp[:, off:off + (n1 - k1)].T @ Pp[:, off:]
        if not two_panel or len(pend) == 2 or n1 == N:
            for (c0, Pp) in pend:
                off = n1 - c0
                A[n1:, n1:] -= Pp[:, off:].T @ Pp[:, off:]
            pend = []
        else:
            pass                                # A step: the rest of the trailing matrix waits for the second panel
    return R


def test_blocked_cholesky_model_needs_the_refinement_step():
    """A kernel matrix of condition ~1e9: the factor from the explicit-inverse panel misses R'R = A by orders of magnitude more
    than a backward-stable factorisation does; one refinement step closes the gap.  (The GPU regression test is
    tests/test_gpu_gplite.py::test_posterior_on_ill_conditioned_kernel_matrices.)"""
    rng = np.random.default_rng(11)
    x = np.sort(rng.uniform(-3, 3, 90))
    K = np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2 / 0.9 ** 2) + 1e-9 * np.eye(90)
    ref = np.linalg.cholesky(K).T
    back = lambda R: np.max(np.abs(R.T @ R - K)) / np.max(np.abs(K))   # noqa: E731  backward error of the factorisation
    e_ref = back(ref)
    for two in (False, True):
        e_plain = back(_blocked_chol_model(K, False, two))
        e_fix = back(_blocked_chol_model(K, True, two))
        assert e_fix < 20 * max(e_ref, 1e-16), (two, e_fix, e_ref)
        assert e_plain > 20 * e_fix, (two, e_plain, e_fix)
    # and the two-panel order is the same factorisation up to rounding
    assert np.max(np.abs(_blocked_chol_model(K, True, True) - _blocked_chol_model(K, True, False))) < 1e-6 * np.max(np.abs(ref))
