"""GPU parity: gplite_post / gplite_pred / sq_dist through the C ABI vs the oracle."""
import math

import numpy as np
import pytest

from oracle import vbmc_ref as R
from tests._cases import golden_cases, load_golden, load_pred_golden, pred_golden_cases, synth_problem
from tests.test_gpu_elbo import relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def va():
    import vbmc_amd

    return vbmc_amd


@pytest.mark.parametrize("shape", [(3, 7, 5), (10, 400, 33), (1, 20, 20), (6, 17, 100), (13, 64, 48)])
def test_sq_dist(va, shape):
    D, n, m = shape
    rng = np.random.default_rng(0)
    a = rng.standard_normal((D, n)) + 50.0  # large common offset: exercises the mean-centring
    b = rng.standard_normal((D, m)) + 50.0
    assert relerr(va.sq_dist(a, b), R.sq_dist(a, b)) < 1e-12
    assert relerr(va.sq_dist(a), R.sq_dist(a)) < 1e-12
    direct = np.sum((a[:, :, None] - b[:, None, :]) ** 2, axis=0)
    assert relerr(va.sq_dist(a, b), direct) < 1e-10
    assert np.all(va.sq_dist(a) >= 0)


@pytest.mark.parametrize("path", golden_cases())
def test_gp_post_pred_golden(va, path):
    inp, exp = load_golden(path)
    gp = va.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, inp["meanfun"])
    for s, post in enumerate(gp["post"]):
        assert post["Lchol"] and post["sn2_mult"] == 1.0
        assert relerr(post["alpha"], exp["alpha"][s]) < 1e-9
        assert relerr(post["L"], exp["L"][s]) < 1e-10
    ymu, ys2, fmu, fs2 = va.gplite_pred(gp, inp["Xstar"], None, None, True)
    assert relerr(np.atleast_2d(fmu.T) if fmu.ndim > 1 else fmu[None, :], np.array(exp["pred_fmu"])) < 1e-9
    assert relerr(np.atleast_2d(fs2.T) if fs2.ndim > 1 else fs2[None, :], np.array(exp["pred_fs2"])) < 1e-9


@pytest.mark.parametrize("cfg", [(6, 200, 8, 4, False), (10, 400, 3, 4, False), (3, 37, 2, 1, False), (5, 90, 3, 4, True), (2, 16, 2, 0, False)])
def test_gp_post_matches_oracle(va, cfg):
    D, N, S, meanfun, noisy = cfg
    p = synth_problem(21, D, N, 3, S, meanfun=meanfun, noisy=noisy)
    ref = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=meanfun, noisefun=p["noisefun"], s2=p["s2"])
    gp = va.gplite_post(p["hyp"], p["X"], p["y"], 1, meanfun, p["noisefun"], p["s2"])
    for a, b in zip(gp["post"], ref["post"]):
        assert a["Lchol"] == b["Lchol"] and a["sn2_mult"] == b["sn2_mult"]
        assert relerr(a["sW"], b["sW"]) < 1e-14
        assert relerr(a["L"], b["L"]) < 1e-9
        assert np.allclose(np.tril(a["L"], -1), 0)
        assert relerr(a["alpha"], b["alpha"]) < 1e-7  # cond(K/sn2 + I) ~ 1e7 at sn2 = 1e-6
        # backward-error check that does not depend on the oracle: (K + sn2 I) alpha = y - m
        hyp = a["hyp"]
        ell = np.exp(hyp[:D])
        Kmat = math.exp(2 * hyp[D]) * np.exp(-0.5 * R.sq_dist(p["X"].T / ell[:, None]))
        sn2 = np.broadcast_to(R.gplite_noisefun(hyp[D + 1:D + 2], p["X"], p["noisefun"], p["y"], p["s2"]), (N,)) * a["sn2_mult"]
        m = R.gplite_meanfun(hyp[D + 2:], p["X"], meanfun)
        res = (Kmat + np.diag(sn2)) @ a["alpha"] - (p["y"] - m)
        assert np.max(np.abs(res)) < 1e-6 * max(1.0, np.max(np.abs(p["y"] - m)))
    Xs = 1.5 * np.random.default_rng(3).standard_normal((70, D))
    s2s = np.full(70, 0.5) if noisy else None
    for ss in (True, False):
        o = va.gplite_pred(gp, Xs, None, s2s, ss)
        r = R.gplite_pred(ref, Xs, None, s2s, ssflag=ss)
        for x, z in zip(o, r):
            assert x.shape == np.asarray(z).shape
            assert relerr(x, z) < 1e-7
    assert np.all(va.gplite_pred(gp, Xs, None, s2s, True)[3] >= 0)


def test_cholesky_jitter_retry_matches_reference_semantics(va):
    """Duplicate inputs + small noise: chol fails, sn2_mult is inflated x10 until it works."""
    rng = np.random.default_rng(0)
    X = rng.standard_normal((12, 2))
    X = np.vstack([X, X, X])
    y = rng.standard_normal(36)
    hyp = np.array([0.0, 0.0, 4.0, math.log(1.05e-3), 0.0, 0.0, 0.0, 0.0, 0.0])[:, None]
    ref = R.gplite_post(hyp, X, y, meanfun=4)
    gp = va.gplite_post(hyp, X, y, 1, 4)
    assert gp["post"][0]["sn2_mult"] == ref["post"][0]["sn2_mult"]
    assert np.all(np.isfinite(gp["post"][0]["alpha"]))
    assert relerr(gp["post"][0]["L"], ref["post"][0]["L"]) < 1e-6


def test_low_noise_branch(va):
    """min(sn2) < 1e-6 -> Lchol = false, L = -inv(K + sn2 I) (gplite_core.m:67,84-99)."""
    p = synth_problem(22, 3, 30, 2, 2)
    hyp = p["hyp"].copy()
    hyp[3 + 1, :] = math.log(3e-4)  # sn2 = 9e-8
    hyp[3, :] -= 1.0
    ref = R.gplite_post(hyp, p["X"], p["y"], meanfun=4)
    gp = va.gplite_post(hyp, p["X"], p["y"], 1, 4)
    for a, b in zip(gp["post"], ref["post"]):
        assert (not a["Lchol"]) and (not b["Lchol"])
        assert relerr(a["L"], b["L"]) < 1e-6
        assert relerr(a["alpha"], b["alpha"]) < 1e-6
    Xs = p["X"][:9] + 0.05
    o = va.gplite_pred(gp, Xs, None, None, False)
    r = R.gplite_pred(ref, Xs)
    assert relerr(o[2], r[2]) < 1e-6 and np.max(np.abs(o[3] - r[3])) < 1e-6


def test_posterior_feeds_elbo_without_reupload(va):
    """gplite_post's device handle is reused by negelcbo (no host round trip of L)."""
    p = synth_problem(23, 4, 60, 5, 3)
    gp = va.gplite_post(p["hyp"], p["X"], p["y"], 1, 4)
    ref = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=4)
    vp = va.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    theta = np.concatenate([p["mu"].reshape(-1, order="F"), np.log(p["sigma"]), np.log(p["lam"]), p["eta"]])
    out = va.negelcbo_vbmc(theta, 0, vp, gp, 0, 0, 1, nargout=11)
    r = R.negelcbo_vbmc(theta, 0, vp, ref, 0, False, 1, separate_K=True)
    assert relerr(out[2], r["G"]) < 1e-7 and relerr(out[7], r["varG"]) < 1e-5


def test_posterior_factors_kept_on_device(va):
    """gplite_post(..., need_L=False): no N x N x S readback; prediction, the full-variance ELBO and a rank-one append all
    read the device copy and agree with the oracle working from its own host factors."""
    p = synth_problem(29, 4, 60, 5, 3)
    gp = va.gplite_post(p["hyp"], p["X"], p["y"], 1, 4, need_L=False)
    ref = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=4)
    assert all(q["L"] is None for q in gp["post"])
    for a, c in zip(gp["post"], ref["post"]):
        assert relerr(a["alpha"], c["alpha"]) < 1e-7 and relerr(a["sW"], c["sW"]) < 1e-12 and a["Lchol"] == c["Lchol"]
    Xq = p["X"][:9] + 0.15
    o = va.gplite_pred(gp, Xq, None, None, True)
    r = R.gplite_pred(ref, Xq, ssflag=True)
    assert relerr(o[2], r[2]) < 1e-7 and relerr(o[3], r[3]) < 1e-6
    vp = va.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    theta = np.concatenate([p["mu"].reshape(-1, order="F"), np.log(p["sigma"]), np.log(p["lam"]), p["eta"]])
    out = va.negelcbo_vbmc(theta, 0, vp, gp, 0, 0, 1, nargout=11)
    rr = R.negelcbo_vbmc(theta, 0, vp, ref, 0, False, 1, separate_K=True)
    assert relerr(out[2], rr["G"]) < 1e-7 and relerr(out[7], rr["varG"]) < 1e-5
    gp1 = va.gplite_post_rank1(gp, Xq[0], 0.4)
    ref1 = R.gplite_post_rank1(ref, Xq[0], 0.4)
    for a, c in zip(gp1["post"], ref1["post"]):
        assert relerr(a["alpha"], c["alpha"]) < 1e-7 and relerr(a["L"], c["L"]) < 1e-8


@pytest.mark.parametrize("seed,D,N", [(0, 1, 19), (0, 1, 40), (5, 2, 70), (0, 1, 150)])
def test_posterior_on_ill_conditioned_kernel_matrices(va, seed, D, N):
    """Kernel matrices of condition 1e7 .. 2e7 (noise-free, one or two dimensions).  The first of them is the example the
    random-shape sweep found at 15x depth when the Cholesky panel became a product with the explicit inverse of the diagonal
    block: alpha was off by 2e-8.  With one step of iterative refinement in the panel the factorisation is as accurate as the
    substitution it replaced."""
    p = synth_problem(seed, D, N, 2, 2, meanfun=0, noisy=False)
    ref = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=0, noisefun=p["noisefun"], s2=p["s2"])
    gp = va.gplite_post(p["hyp"], p["X"], p["y"], 1, 0, p["noisefun"], p["s2"])
    for a, b in zip(gp["post"], ref["post"]):
        assert a["Lchol"] == b["Lchol"] and a["sn2_mult"] == b["sn2_mult"]
        assert relerr(a["alpha"], b["alpha"]) < 1e-8
        assert relerr(a["L"], b["L"]) < 1e-10


@pytest.mark.parametrize("N", [497, 512, 513, 560, 561, 592, 593, 1104, 1105])
def test_posterior_at_the_kernel_switch_points(va, N):
    """N = 512 | 513: the single right-hand-side solve changes kernels (k_alpha_solve1 keeps one vector element per thread of a
    512-thread workgroup); N = 560 | 561: the Cholesky kernel (with the right-hand side riding along: Np more doubles of LDS) drops
    from two panels in LDS to one (592 | 593 before round 5); 1104 | 1105: the panel moves to a global scratch block; 497: a ragged
    last block."""
    p = synth_problem(40 + N, 3, N, 2, 2, meanfun=4, noisy=True)      # condition ~1e3: the tolerances test the kernels, not the problem
    ref = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=4, noisefun=p["noisefun"], s2=p["s2"])
    gp = va.gplite_post(p["hyp"], p["X"], p["y"], 1, 4, p["noisefun"], p["s2"])
    for a, b in zip(gp["post"], ref["post"]):
        assert a["Lchol"] == b["Lchol"] and a["sn2_mult"] == b["sn2_mult"]
        assert relerr(a["alpha"], b["alpha"]) < 1e-9 and relerr(a["L"], b["L"]) < 1e-11
    Xq = p["X"][:6] + 0.05
    o = va.gplite_pred(gp, Xq, None, None, True)
    r = R.gplite_pred(ref, Xq, ssflag=True)
    assert relerr(o[2], r[2]) < 1e-7 and np.max(np.abs(o[3] - r[3])) < 1e-6 * max(1.0, np.max(np.abs(r[3])))


def test_posterior_when_pivots_leave_the_fast_tile_range(va):
    """The Cholesky kernel's fast diagonal-tile routine (blocked LDL', chol_mfma.h: chol_diag_tile_fast) declines a tile with a pivot
    outside (2^-200, 2^200) and the careful pivot-by-pivot routine redoes it: a signal variance of e^160 over a noise of e^-6 puts
    the pivots of K / sn2 + I near 2^240 -- the posterior must still match the oracle (and nothing may be flagged as not positive
    definite); the same with a tiny signal (pivots stay ~1: the fast path) as the control."""
    for lnsf in (80.0, -3.0):
        p = synth_problem(7, 3, 53, 2, 2, meanfun=1, noisy=False)
        p["hyp"][3, :] = lnsf                       # ln sf
        p["hyp"][4, :] = -3.0                       # ln sn
        ref = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=1, noisefun=p["noisefun"], s2=p["s2"])
        gp = va.gplite_post(p["hyp"], p["X"], p["y"], 1, 1, p["noisefun"], p["s2"])
        for a, b in zip(gp["post"], ref["post"]):
            assert a["Lchol"] == b["Lchol"] and a["sn2_mult"] == b["sn2_mult"] == 1.0
            assert relerr(a["L"], b["L"]) < 1e-10, (lnsf, relerr(a["L"], b["L"]))
            assert relerr(a["alpha"], b["alpha"]) < 1e-6, (lnsf, relerr(a["alpha"], b["alpha"]))


def test_rank1_update_equals_full_posterior(va):
    """gplite/gplite_test.m:87-105 property: appending a point by the rank-1 path == full recompute."""
    p = synth_problem(24, 4, 45, 3, 3)
    gp_small = va.gplite_post(p["hyp"], p["X"][:-1], p["y"][:-1], 1, 4)
    gp_r1 = va.gplite_post_rank1(gp_small, p["X"][-1], p["y"][-1])
    gp_full = va.gplite_post(p["hyp"], p["X"], p["y"], 1, 4)
    ref_r1 = R.gplite_post_rank1(R.gplite_post(p["hyp"], p["X"][:-1], p["y"][:-1], meanfun=4), p["X"][-1], p["y"][-1])
    for a, b, c in zip(gp_r1["post"], gp_full["post"], ref_r1["post"]):
        assert a["alpha"].shape == (45,) and a["L"].shape == (45, 45) and a["sW"].shape == (45,)
        assert relerr(a["alpha"], c["alpha"]) < 1e-7 and relerr(a["L"], c["L"]) < 1e-8  # vs the oracle's rank-1
        assert relerr(a["alpha"], b["alpha"]) < 1e-6 and relerr(a["L"], b["L"]) < 1e-7  # vs the full update
    # the enlarged posterior is usable downstream
    o = va.gplite_pred(gp_r1, p["X"][:5] + 0.1, None, None, False)
    r = R.gplite_pred(R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=4), p["X"][:5] + 0.1)
    assert relerr(o[2], r[2]) < 1e-6


def test_rank1_chain_on_device_and_low_noise_branch(va):
    """Three appends in a row with the factors kept on the device only (need_L=False), then a prediction from the device
    copy, against the oracle's chained rank-1 updates; and one append on the low-noise branch, where gp.post(s).L is
    -inv(K + sn2 I) and the update is the bordered-inverse formula (gplite_post.m:234-236)."""
    p = synth_problem(31, 3, 40, 3, 2)
    n0 = 37
    gp = va.gplite_post(p["hyp"], p["X"][:n0], p["y"][:n0], 1, 4)
    ref = R.gplite_post(p["hyp"], p["X"][:n0], p["y"][:n0], meanfun=4)
    for i in range(n0, 40):
        gp = va.gplite_post_rank1(gp, p["X"][i], p["y"][i], need_L=False)
        ref = R.gplite_post_rank1(ref, p["X"][i], p["y"][i])
        assert gp["post"][0]["L"] is None and gp["X"].shape == (i + 1, 3)
        for a, c in zip(gp["post"], ref["post"]):
            assert relerr(a["alpha"], c["alpha"]) < 1e-7 and relerr(a["sW"], c["sW"]) < 1e-12
    Xq = p["X"][:7] + 0.2
    o = va.gplite_pred(gp, Xq, None, None, True)
    r = R.gplite_pred(ref, Xq, ssflag=True)
    assert relerr(o[2], r[2]) < 1e-7 and relerr(o[3], r[3]) < 1e-6
    # the last append with L read back equals the oracle's matrix
    gpL = va.gplite_post_rank1(va.gplite_post(p["hyp"], p["X"][:39], p["y"][:39], 1, 4), p["X"][39], p["y"][39])
    refL = R.gplite_post_rank1(R.gplite_post(p["hyp"], p["X"][:39], p["y"][:39], meanfun=4), p["X"][39], p["y"][39])
    for a, c in zip(gpL["post"], refL["post"]):
        assert relerr(a["L"], c["L"]) < 1e-8
    # low-noise branch
    hyp = p["hyp"].copy()
    hyp[4, :] = np.log(3e-4)          # sn2 = 9e-8 < 1e-6 -> Lchol = false (gplite_core.m:67)
    g0 = va.gplite_post(hyp, p["X"][:39], p["y"][:39], 1, 4)
    r0 = R.gplite_post(hyp, p["X"][:39], p["y"][:39], meanfun=4)
    assert not g0["post"][0]["Lchol"]
    g1 = va.gplite_post_rank1(g0, p["X"][39], p["y"][39])
    r1 = R.gplite_post_rank1(r0, p["X"][39], p["y"][39])
    for a, c in zip(g1["post"], r1["post"]):
        scale = np.max(np.abs(c["L"]))
        assert np.max(np.abs(a["L"] - c["L"])) < 1e-6 * scale and relerr(a["alpha"], c["alpha"]) < 1e-5


@pytest.mark.parametrize("N", [39, 1070])
def test_low_noise_posterior_inverse(va, N):
    """Low-noise samples store L = -inv(K + sn2 I) (gplite_core.m:84,98), formed by k_spd_inverse: paired column blocks with
    an odd block count (N = 39), and N = 1070 where the two slabs no longer share the LDS (one column block per workgroup)."""
    rng = np.random.default_rng(N)
    D = 3
    X = 1.5 * rng.standard_normal((N, D))
    y = -0.5 * np.sum(X ** 2, axis=1) + 0.1 * rng.standard_normal(N)
    hyp = np.zeros((D + 2 + 2 * D + 1, 1))
    hyp[:D, 0] = np.log(0.25)                  # short length scales keep K + sn2 I well conditioned at sn2 = 9e-8
    hyp[D, 0] = np.log(np.std(y))
    hyp[D + 1, 0] = np.log(3e-4)
    hyp[D + 2, 0] = np.max(y)
    hyp[D + 3 + D:, 0] = np.log(2.0)
    gp = va.gplite_post(hyp, X, y, 1, 4)
    ref = R.gplite_post(hyp, X, y, meanfun=4)
    assert not gp["post"][0]["Lchol"] and not ref["post"][0]["Lchol"]
    Lr = ref["post"][0]["L"]
    assert np.max(np.abs(gp["post"][0]["L"] - Lr)) < 1e-7 * np.max(np.abs(Lr))
    assert relerr(gp["post"][0]["alpha"], ref["post"][0]["alpha"]) < 1e-6


def test_pred_and_acq_chunking_over_many_points(va):
    """Sweeps whose S x N x Nstar cross-kernel matrix would exceed 1 GiB are cut into chunks of test points; the
    chunked result equals the one-shot result up to the sq_dist centring constant (rounding)."""
    from tests.test_gpu_elbo import problem
    p, gp, vp, _ = problem(61, 6, 300, 5, 24)          # S*N*8 = 57.6 KB per point -> chunks of 18640 points
    rng = np.random.default_rng(0)
    Xs = 1.3 * rng.standard_normal((40000, 6))
    fmu, fs2 = va.gplite_pred(gp, Xs, None, None, True)[2:4]
    idx = rng.choice(40000, 600, replace=False)
    fmu_s, fs2_s = va.gplite_pred(gp, Xs[idx], None, None, True)[2:4]
    sf2 = np.exp(2 * gp["post"][0]["hyp"][6])
    assert relerr(np.asarray(fmu)[idx], fmu_s) < 1e-10 and np.max(np.abs(np.asarray(fs2)[idx] - np.asarray(fs2_s))) < 1e-10 * sf2
    st = {"ymax": float(np.max(gp["y"])), "VarianceRegularizedAcqFcn": False, "TolGPVar": 1e-4}
    acq = va.acqwrapper_vbmc(Xs, vp, gp, st, False, "acqflog_vbmc", None)
    acq_s = va.acqwrapper_vbmc(Xs[idx], vp, gp, st, False, "acqflog_vbmc", None)
    far = np.asarray(fs2_s).mean(axis=1) > 1e-6 * sf2
    assert np.max(np.abs(acq[idx][far] - acq_s[far])) < 1e-7


@pytest.mark.parametrize("cfg", [(4, 60, 3, 3000), (7, 130, 1, 2500), (3, 17, 2, 1111)])
def test_pred_many_points_few_samples(va, cfg):
    """Few hyper-samples and many test points: the point tiles are sliced over gridDim.z so that the chip is covered."""
    from tests.test_gpu_elbo import problem
    D, N, S, nstar = cfg
    p, gp, vp, _ = problem(71, D, N, 3, S)
    Xs = 1.4 * np.random.default_rng(3).standard_normal((nstar, D))
    r_d = va.gplite_pred(gp, Xs, None, None, True)
    r_o = R.gplite_pred(gp, Xs, None, None, True)
    sf2 = np.exp(2 * gp["post"][0]["hyp"][D])
    assert relerr(np.asarray(r_d[2]).reshape(-1), np.asarray(r_o[2]).reshape(-1)) < 1e-9
    assert np.max(np.abs(np.asarray(r_d[3]).reshape(-1) - np.asarray(r_o[3]).reshape(-1))) < 1e-9 * sf2
    avg_d = va.gplite_pred(gp, Xs, None, None, False)        # hyper-sample average + between-sample variance (:154-165)
    avg_o = R.gplite_pred(gp, Xs, None, None, False)
    assert relerr(avg_d[0], avg_o[0]) < 1e-9 and np.max(np.abs(np.asarray(avg_d[1]) - np.asarray(avg_o[1]))) < 1e-9 * sf2


def test_pred_log_predictive_density(va):
    """lp (5th output of gplite_pred, :124-127): per hyper-sample, also when the other outputs are averaged."""
    from tests.test_gpu_elbo import problem
    p, gp, vp, _ = problem(81, 4, 50, 3, 3)
    rng = np.random.default_rng(1)
    Xs = 1.2 * rng.standard_normal((40, 4))
    ys = rng.standard_normal(40) - 3.0
    for ss in (True, False):
        d = va.gplite_pred(gp, Xs, ys, None, ss, nargout=5)
        o = R.gplite_pred(gp, Xs, ys, None, ss, nargout=5)
        assert d[4].shape == (40, 3) and relerr(d[4], o[4]) < 1e-8
        assert relerr(d[0], o[0]) < 1e-9 and np.asarray(d[0]).shape == np.asarray(o[0]).shape
    assert va.gplite_pred(gp, Xs, None, None, True, nargout=5)[4] is None


@pytest.mark.parametrize("path", pred_golden_cases())
def test_pred_general_noise_golden(va, path):
    """Device gplite_post + gplite_pred(gp,Xstar,ystar,s2star,1) with every noise model of gplite_noisefun.m:176-210
    (incl. the output-dependent term, which needs ystar at the test points) and lp, on both Lchol branches, against the
    50-digit vectors tests/golden/mp_pred_case*.json."""
    from tests.test_oracle_golden import check_pred_against_golden
    inp, exp = load_pred_golden(path)
    gp = va.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, inp["meanfun"], inp["noisefun"], inp["s2"])
    out = va.gplite_pred(gp, inp["Xstar"], inp["ystar"], inp["s2star"], True, nargout=5)
    check_pred_against_golden(gp, out, exp, bool(np.min(exp["min_sn2"]) >= 1e-6))
    # averaged outputs (gplite_pred.m:154-165) from the per-sample vectors
    S = exp["fmu"].shape[0]
    if S > 1:
        ymu, ys2, fmu, fs2 = va.gplite_pred(gp, inp["Xstar"], inp["ystar"], inp["s2star"], False)
        fbar = np.mean(exp["fmu"], axis=0)
        assert relerr(fmu, fbar) < 1e-8 and relerr(ymu, fbar) < 1e-8
        assert relerr(fs2, np.mean(exp["fs2"], axis=0) + np.var(exp["fmu"], axis=0, ddof=1)) < 1e-8
        assert relerr(ys2, np.mean(exp["ys2"], axis=0) + np.var(exp["fmu"], axis=0, ddof=1)) < 1e-8
