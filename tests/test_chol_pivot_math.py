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


# ------------------------------------------------------------------------------------------
# Round 5: the diagonal tile as a blocked LDL' (chol_mfma.h: chol_diag_tile_fast) -- four columns at a time, the 4 x 4 diagonal
# block eliminated on wave-uniform values with refined reciprocals (v_rcp_f64 + two Newton steps), the rows of the tile
# against it, the rank-4 update as one MFMA, L = (L~ D) D^-1/2 at the end; the inverse of the factor by a blocked
# substitution on the UNIT factor L~ (cross-block sums split over the four lane groups, the 4 x 4 diagonal coupling with
# uniform coefficients), rows scaled by D^-1/2.  NumPy model of exactly that algebra.
def _rcp2(d, seed_rel_err=2.0 ** -26):
    y = (1.0 / d) * (1.0 + seed_rel_err)                  # stand-in for v_rcp_f64
    for _ in range(2):
        e = fma(-d, y, 1.0)
        y = fma(y, e, y)
    return y


def _tile_ldl_model(T):
    """returns (L, W): lower Cholesky factor of the 16 x 16 SPD tile T and W = inv(L), computed the way the kernel does"""
    T = np.array(T, dtype=np.float64)
    n = 16
    Lt = np.zeros((n, n))          # unit lower factor L~ (strictly lower part)
    U = np.zeros((n, n))           # U[i][s] = d_s * L~[i][s]
    d = np.zeros(n)
    for b in range(4):
        c0 = 4 * b
        a = T[c0:c0 + 4, c0:c0 + 4]
        # uniform 4 x 4 LDL'
        d0 = a[0, 0]; y0 = _rcp2(d0)
        l10, l20, l30 = a[1, 0] * y0, a[2, 0] * y0, a[3, 0] * y0
        d1 = fma(-l10, a[1, 0], a[1, 1]); y1 = _rcp2(d1)
        u21 = fma(-l20, a[1, 0], a[2, 1]); u31 = fma(-l30, a[1, 0], a[3, 1])
        l21, l31 = u21 * y1, u31 * y1
        d2 = fma(-l21, u21, fma(-l20, a[2, 0], a[2, 2])); y2 = _rcp2(d2)
        u32 = fma(-l31, u21, fma(-l30, a[2, 0], a[3, 2])); l32 = u32 * y2
        d3 = fma(-l32, u32, fma(-l31, u31, fma(-l30, a[3, 0], a[3, 3]))); y3 = _rcp2(d3)
        d[c0:c0 + 4] = d0, d1, d2, d3
        # every row against the block
        for i in range(c0, n):
            w = T[i, c0:c0 + 4]
            u0 = w[0]; l0 = u0 * y0
            u1 = fma(-l0, a[1, 0], w[1]); l1 = u1 * y1
            u2 = fma(-l1, u21, fma(-l0, a[2, 0], w[2])); l2 = u2 * y2
            u3 = fma(-l2, u32, fma(-l1, u31, fma(-l0, a[3, 0], w[3]))); l3 = u3 * y3
            U[i, c0:c0 + 4] = u0, u1, u2, u3
            Lt[i, c0:c0 + 4] = l0, l1, l2, l3
        # rank-4 update of the columns to the right (lower triangle)
        for i in range(c0 + 4, n):
            for j in range(c0 + 4, i + 1):
                T[i, j] -= float(np.dot(Lt[i, c0:c0 + 4], U[j, c0:c0 + 4]))
    ri = np.array([chol_sqrt_rsqrt(float(x), 2.0 ** -21)[1] for x in d])
    rs = np.array([chol_sqrt_rsqrt(float(x), 2.0 ** -21)[0] for x in d])
    L = np.tril(U * ri[None, :], -1) + np.diag(rs)
    # inverse of the unit factor, blocks of four rows
    M = np.zeros((n, n))
    for c in range(n):
        r = np.zeros(n)
        for k in range(4):
            s = np.zeros(4)
            for g in range(4):                      # the four lane groups: u = 4 kk + g
                part = np.zeros(4)
                for kk in range(k):
                    u = 4 * kk + g
                    for p in range(4):
                        part[p] = fma(Lt[4 * k + p, u], r[u], part[p])
                s += part
            bvec = [(1.0 if c == 4 * k + p else 0.0) - s[p] for p in range(4)]
            q = 4 * k
            r0 = bvec[0]
            r1 = fma(-Lt[q + 1, q], r0, bvec[1])
            r2 = fma(-Lt[q + 2, q + 1], r1, fma(-Lt[q + 2, q], r0, bvec[2]))
            r3 = fma(-Lt[q + 3, q + 2], r2, fma(-Lt[q + 3, q + 1], r1, fma(-Lt[q + 3, q], r0, bvec[3])))
            r[q:q + 4] = r0, r1, r2, r3
        M[:, c] = r
    W = ri[:, None] * M
    return L, W


def test_blocked_ldl_tile_model_factor_and_inverse():
    rng = np.random.default_rng(11)
    for trial in range(6):
        G = rng.standard_normal((16, 40 if trial < 3 else 17))
        T = G @ G.T / G.shape[1] + (0.05 if trial % 2 else 1e-3) * np.eye(16)
        L, W = _tile_ldl_model(T)
        Lref = np.linalg.cholesky(T)
        scale = np.abs(Lref).max()
        cond = np.linalg.cond(Lref)
        assert np.abs(L - Lref).max() < 40 * cond * np.finfo(float).eps * scale, (trial, np.abs(L - Lref).max(), cond)
        assert np.abs(L @ L.T - T).max() < 64 * np.finfo(float).eps * np.abs(T).max()
        assert np.abs(W @ L - np.eye(16)).max() < 64 * cond * np.finfo(float).eps
        assert np.abs(np.triu(W, 1)).max() == 0.0
