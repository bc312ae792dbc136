"""GPU parity of the small-class entropy kernel (vbmc_amd/csrc/entropy_lane.h: K <= 16, D <= 12, one lane per base sample, the
expected log joint as a role of the same launch) against the oracle AND against the matrix-core kernel on the same inputs and the
same random stream, over the edges of the class: K = 1 and 16, D = 1 and 12, sample counts that are not multiples of the 64-lane
tile (and smaller than one), parity mode (the caller's draws), value-only calls, the role switched off, a training set too large for
the role's LDS block, restarts dealt by key.  Tolerances as tests/test_gpu_elbo.py: 1e-10 values, 1e-9 gradients (fp64 end to end).
"""
import os

import numpy as np
import pytest

from oracle import vbmc_ref as R
from tests._cases import relerr, synth_problem

pytestmark = pytest.mark.gpu

RT_VAL = 1e-10
RT_GRAD = 1e-9


@pytest.fixture(scope="module")
def va():
    import vbmc_amd

    return vbmc_amd


def problem(seed, D, N, K, S, **kw):
    p = synth_problem(seed, D, N, K, S, **kw)
    gp = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=p["meanfun"], noisefun=p["noisefun"], s2=p["s2"])
    vp = R.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    theta = np.concatenate([p["mu"].reshape(-1, order="F"), np.log(p["sigma"]), np.log(p["lam"]), p["eta"]])
    return p, gp, vp, theta


class env:
    """the library reads its A/B switches with getenv at plan time: set for the duration of a call"""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_the_plan_hook_names_the_class(va):
    import ctypes

    from vbmc_amd import _lib

    lib = _lib.load()
    q = [ctypes.c_int() for _ in range(4)]
    for D, K, kind in ((6, 10, 2), (1, 1, 2), (8, 16, 2), (12, 8, 2), (10, 10, 2), (12, 9, 1), (10, 11, 1), (12, 16, 1), (13, 16, 1), (12, 17, 1), (6, 50, 1)):
        assert lib.vbmc_entropy_plan(D, K, *[ctypes.byref(x) for x in q]) == kind, (D, K)
    assert lib.vbmc_entropy_plan(5, 7, *[ctypes.byref(x) for x in q]) == 2 and (q[0].value, q[1].value, q[2].value) == (6, 8, 4)


SHAPES = [
    # D, N, K, S, Ns
    (6, 200, 10, 8, 1000),      # BASELINE configs[1]
    (1, 20, 1, 1, 10),          # one component: q = its own density, H = its entropy
    (1, 30, 16, 2, 130),
    (8, 50, 16, 2, 200),        # the corners of the class
    (12, 50, 8, 2, 200),
    (10, 50, 10, 2, 300),
    (12, 40, 1, 1, 64),
    (2, 30, 2, 1, 100),         # BASELINE configs[0]
    (5, 33, 7, 3, 2),           # one antithetic pair per component
    (3, 17, 5, 2, 37),          # Mh = 19 < 64: one partial tile
    (7, 64, 9, 2, 129),         # Mh = 65: a full tile and one lane
    (11, 45, 13, 2, 1500),
    (4, 600, 3, 2, 90),         # N = 600 at S = 2: X (4 x 640) + 4 alpha blocks fit the role's 48 KB
    (8, 700, 6, 3, 70),         # ... and here they do not: the separate log-joint kernel
]


@pytest.mark.parametrize("D,N,K,S,Ns", SHAPES)
def test_lane_kernel_matches_oracle_and_matrix_core_kernel(va, monkeypatch, D, N, K, S, Ns):
    monkeypatch.setenv("VBMC_ENT_KERNEL", "lane")    # (a single evaluation is below the width the policy gives the lane kernel: ask for it)
    p, gp, vp, theta = problem(100 + D + K, D, N, K, S)
    Mh = (Ns + 1) // 2
    eps = np.random.default_rng(5).standard_normal((K, Mh, D))
    ref = R.negelcbo_vbmc(theta, 0, vp, gp, Ns, True, 0, eps=eps)
    r = va.negelcbo_batch(theta, 0, vp, gp, Ns, True, 0, eps=eps)
    assert relerr(r["F"][0], ref["F"]) < RT_VAL and relerr(r["H"][0], ref["H"]) < RT_VAL and relerr(r["G"][0], ref["G"]) < RT_VAL
    assert relerr(r["dF"][:, 0], ref["dF"]) < RT_GRAD and relerr(r["dH"][:, 0], ref["dH"]) < RT_GRAD
    assert relerr(r["dG"][:, 0], ref["dG"]) < RT_GRAD
    # the same call on the matrix-core kernel, and the device RNG stream through both
    with env(VBMC_ENT_KERNEL="mfma"):
        m = va.negelcbo_batch(theta, 0, vp, gp, Ns, True, 0, eps=eps)
        md = va.negelcbo_batch(theta, 0, vp, gp, Ns, True, 0, seed=11)
    ld = va.negelcbo_batch(theta, 0, vp, gp, Ns, True, 0, seed=11)
    for a, b in ((r, m), (ld, md)):
        assert relerr(a["H"][0], b["H"][0]) < 1e-12 and relerr(a["dH"][:, 0], b["dH"][:, 0]) < 1e-11
        # (the expected log joint is a sum of terms z_n alpha_n that cancel: both are compared with the oracle at 1e-10 above; against
        # each other the two exponentials -- 256- and 1024-entry tables -- show through that cancellation)
        assert relerr(a["G"][0], b["G"][0]) < RT_VAL and relerr(a["dG"][:, 0], b["dG"][:, 0]) < RT_GRAD
    # value only; the role switched off (separate log-joint kernel): the entropy part does not move at all
    v = va.negelcbo_batch(theta, 0, vp, gp, Ns, False, 0, seed=11)
    assert relerr(v["H"][0], ld["H"][0]) < 1e-13 and relerr(v["G"][0], ld["G"][0]) < RT_VAL
    with env(VBMC_LJ_CO="0"):
        o = va.negelcbo_batch(theta, 0, vp, gp, Ns, True, 0, seed=11)
    assert np.array_equal(o["H"], ld["H"]) and np.array_equal(o["dH"], ld["dH"])
    assert relerr(o["G"][0], ld["G"][0]) < RT_VAL and relerr(o["dG"][:, 0], ld["dG"][:, 0]) < RT_GRAD


def test_the_policy_gives_wide_batches_to_the_lane_kernel(va):
    """K R tiles >= 96: the lane kernel; a single chain: the matrix-core kernel (its role splits the training set over more waves) --
    seen from outside as bit-identity with the forced choice"""
    D, N, K, S, Ns = 6, 80, 10, 3, 300
    p, gp, vp, theta = problem(7, D, N, K, S)
    th = np.asfortranarray(theta[:, None] + 0.05 * np.random.default_rng(3).standard_normal((theta.size, 4)))
    wide = va.negelcbo_batch(th, 0, vp, gp, Ns, True, 0, seed=21)           # 10 x 4 x 3 = 120 tiles
    one = va.negelcbo_batch(th[:, 0], 0, vp, gp, Ns, True, 0, seed=21)      # 30 tiles
    with env(VBMC_ENT_KERNEL="lane"):
        wl = va.negelcbo_batch(th, 0, vp, gp, Ns, True, 0, seed=21)
    with env(VBMC_ENT_KERNEL="mfma"):
        om = va.negelcbo_batch(th[:, 0], 0, vp, gp, Ns, True, 0, seed=21)
        wm = va.negelcbo_batch(th, 0, vp, gp, Ns, True, 0, seed=21)
    assert np.array_equal(wide["dF"], wl["dF"]) and np.array_equal(one["dF"], om["dF"])
    assert not np.array_equal(wide["dH"], wm["dH"])        # (the two kernels sum in different orders: equal to rounding, not to the bit)
    assert relerr(wide["dH"], wm["dH"]) < 1e-12


def test_batch_of_restarts_and_restart_keys(va, monkeypatch):
    """R = 5 restarts in one launch: every column equals the same theta evaluated alone under its restart key, bit for bit (a restart's
    records do not depend on the batch it sits in), and matches the oracle on the dumped device stream"""
    monkeypatch.setenv("VBMC_ENT_KERNEL", "lane")
    D, N, K, S, Ns, Rn = 6, 80, 10, 3, 300, 5
    p, gp, vp, theta = problem(7, D, N, K, S)
    th = np.asfortranarray(theta[:, None] + 0.05 * np.random.default_rng(3).standard_normal((theta.size, Rn)))
    b = va.negelcbo_batch(th, 0, vp, gp, Ns, True, 0, seed=21)
    for r in range(Rn):
        one = va.negelcbo_batch(th[:, r], 0, vp, gp, Ns, True, 0, seed=21, restart_offset=r)
        assert np.array_equal(one["H"][0], b["H"][r]) and np.array_equal(one["dH"][:, 0], b["dH"][:, r])
        assert relerr(one["G"][0], b["G"][r]) < 1e-13 and relerr(one["dG"][:, 0], b["dG"][:, r]) < 1e-12
    eps = va.default_engine().ctx.rng_dump(D, K, Rn, Ns, 21)
    for r in (0, Rn - 1):
        ref = R.negelcbo_vbmc(th[:, r], 0, vp, gp, Ns, True, 0, eps=eps[r])
        assert relerr(b["F"][r], ref["F"]) < RT_VAL and relerr(b["dF"][:, r], ref["dF"]) < RT_GRAD


def test_chunking_does_not_change_the_sum_beyond_rounding(va, monkeypatch):
    monkeypatch.setenv("VBMC_ENT_KERNEL", "lane")
    D, N, K, S, Ns = 6, 60, 10, 2, 2000
    p, gp, vp, theta = problem(9, D, N, K, S)
    base = va.negelcbo_batch(theta, 0, vp, gp, Ns, True, 0, seed=4)
    for c in ("1", "2", "5", "16"):
        with env(VBMC_ENT_CHUNKS=c):
            o = va.negelcbo_batch(theta, 0, vp, gp, Ns, True, 0, seed=4)
        assert relerr(o["H"][0], base["H"][0]) < 1e-13 and relerr(o["dH"][:, 0], base["dH"][:, 0]) < 1e-12
