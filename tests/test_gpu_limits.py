"""GPU: shapes beyond the round-1 limits against the oracle -- the deterministic entropy bound with
K > 128 (its K x K table no longer has to fit the LDS) and mixtures whose finalize record exceeds the LDS
(4 D K + 9 K > 19400, e.g. D = 32 with K > 141)."""
import numpy as np
import pytest

from oracle import vbmc_ref as R
from tests._cases import block_relerr, synth_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def va():
    import vbmc_amd

    return vbmc_amd


def _problem(seed, D, N, K, S):
    p = synth_problem(seed, D, N, K, S)
    gp = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=4)
    vp = R.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    theta = np.concatenate([p["mu"].reshape(-1, order="F"), np.log(p["sigma"]), np.log(p["lam"]), p["eta"]])
    _, tb = R.vpbounds(vp, gp, dict(TolLength=1e-6, TolWeight=1e-2, TolConLoss=0.01, WeightPenalty=0.1))
    return gp, vp, theta, tb


@pytest.mark.parametrize("cfg", [(5, 60, 200, 2), (3, 40, 129, 2), (8, 50, 256, 1)])
def test_entlb_beyond_128_components(va, cfg):
    D, N, K, S = cfg
    gp, vp, theta, tb = _problem(71, D, N, K, S)
    ref = R.negelcbo_vbmc(theta, 0, vp, gp, 0, True, 0, thetabnd=tb)
    got = va.negelcbo_batch(np.stack([theta, theta + 0.01], axis=1), 0, vp, gp, 0, True, 0, tb)
    assert abs(got["F"][0] - ref["F"]) < 1e-10 * max(1.0, abs(ref["F"])) and abs(got["H"][0] - ref["H"]) < 1e-10 * max(1.0, abs(ref["H"]))
    assert all(v < 1e-9 for v in block_relerr(got["dF"][:, 0], ref["dF"], D, K).values())
    assert all(v < 1e-9 for v in block_relerr(got["dH"][:, 0], ref["dH"], D, K).values())
    H, dH = va.entlb_vbmc(vp, None, True)
    Hr, dHr = R.entlb_vbmc(vp, grad_flags=True)
    assert abs(H - Hr) < 1e-10 * max(1.0, abs(Hr))


# Round 5: K = 300, 384, 400 -- beyond the 256 of four waves x 64 components: eight-wave workgroups (HV = 8, three or
# four k-tiles per wave), Monte-Carlo entropy and the deterministic bound, with bounds and penalties
@pytest.mark.parametrize("cfg", [(32, 40, 160, 2, 24), (32, 40, 150, 1, 0), (24, 30, 250, 1, 0), (28, 30, 190, 1, 12), (24, 30, 250, 1, 20),
                                 (5, 40, 200, 2, 64), (10, 40, 129, 1, 40), (6, 30, 256, 1, 32),
                                 (10, 40, 300, 2, 40), (4, 30, 384, 1, 32), (6, 30, 400, 1, 24), (10, 40, 300, 1, 0), (18, 30, 300, 1, 20)])
def test_mixtures_whose_finalize_record_exceeds_the_lds(va, cfg):
    D, N, K, S, Ns = cfg
    assert 4 * D * K + 9 * K > 19400 or K > 128      # the old finalize-record limit, or the four- / eight-wave MFMA entropy kernels (K > 128)
    gp, vp, theta, tb = _problem(72, D, N, K, S)
    eps = np.random.default_rng(4).standard_normal((K, max(Ns, 2) // 2, D)) if Ns else None
    ref = R.negelcbo_vbmc(theta, 0, vp, gp, Ns, True, 0, thetabnd=tb, eps=eps)
    got = va.negelcbo_batch(theta, 0, vp, gp, Ns, True, 0, tb, eps=eps)
    assert abs(got["F"][0] - ref["F"]) < 1e-10 * max(1.0, abs(ref["F"]))
    assert abs(got["G"][0] - ref["G"]) < 1e-10 * max(1.0, abs(ref["G"])) and abs(got["H"][0] - ref["H"]) < 1e-10 * max(1.0, abs(ref["H"]))
    for key in ("dF", "dG", "dH"):
        assert all(v < 1e-9 for v in block_relerr(got[key][:, 0], ref[key], D, K).values()), key


# ---- N beyond the 16-column right-hand-side slab (N > 1136): narrow slabs (8 columns up to N = 2144, 4 up to 3872), the
# Cholesky panel in global scratch (N > 1232), slab-solve prediction (N > 1248).  VBMC's default MaxFunEvals = 50 (2 + D)
# reaches N = 1150 at D = 21 and 1700 at D = 32.
def _big_gp(seed, D, N, S, noisy=False, low_noise=False):
    p = synth_problem(seed, D, N, 4, S, noisy=noisy)
    if low_noise:
        p["hyp"][D + 1, :] = np.log(3e-4)      # sn2 = 9e-8 < 1e-6: the stored -inv(K + sn2 I) branch (gplite_core.m:84)
    return p


# Round 5: N = 4500 -- beyond the 3872 of the four-column slab: two right-hand sides per wave (up to N = 6800; one up
# to 9696), every GP / variance path.
@pytest.mark.parametrize("cfg", [(6, 1500, 2, False), (4, 2200, 1, False), (5, 1300, 2, True), (3, 4500, 1, False)])
def test_gp_post_pred_rank1_beyond_the_wide_slab(va, cfg):
    D, N, S, low = cfg
    p = _big_gp(81, D, N, S, low_noise=low)
    ref = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=4)
    gp = va.gplite_post(p["hyp"], p["X"], p["y"], 1, 4, (1, 0, 0), None)
    for a, b in zip(gp["post"], ref["post"]):
        assert a["Lchol"] == b["Lchol"] == (not low) and a["sn2_mult"] == b["sn2_mult"]
        sc = np.max(np.abs(b["L"]))
        assert np.max(np.abs(a["L"] - b["L"])) < (1e-9 if not low else 1e-6) * sc
        assert np.max(np.abs(a["alpha"] - b["alpha"])) < 1e-6 * np.max(np.abs(b["alpha"]))
    Xs = 1.4 * np.random.default_rng(1).standard_normal((70, D))
    o = va.gplite_pred(gp, Xs, None, None, True)
    r = R.gplite_pred(ref, Xs, None, None, ssflag=True)
    sf2 = np.exp(2 * p["hyp"][D, 0])
    flat = lambda z: np.asarray(z, dtype=np.float64).reshape(-1, order="F")   # noqa: E731  (S = 1: (Nstar,) vs (Nstar, 1))
    assert np.max(np.abs(flat(o[2]) - flat(r[2]))) < 1e-6 * max(1.0, np.max(np.abs(r[2])))
    assert np.max(np.abs(flat(o[3]) - flat(r[3]))) < 1e-7 * sf2
    # acquisition sweep on the same surrogate (prediction fused behind it)
    vp = R.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    st = {"ymax": float(np.max(p["y"])), "VarianceRegularizedAcqFcn": False, "TolGPVar": 1e-4}
    acq = va.acqwrapper_vbmc(Xs, vp, gp, st, False, "acqf_vbmc", None)
    acr = R.acqwrapper_vbmc(Xs, vp, ref, st, "acqf")[0]
    assert np.max(np.abs(acq - acr)) < 1e-6 * max(1e-300, np.max(np.abs(acr)))
    if not low:
        # rank-one append == full update with the extra point
        xs, ys = 0.3 * np.ones(D), float(np.mean(p["y"]))
        g1 = va.gplite_post_rank1(gp, xs, ys)
        g2 = R.gplite_post(p["hyp"], np.vstack([p["X"], xs[None, :]]), np.concatenate([p["y"], [ys]]), meanfun=4)
        for a, b in zip(g1["post"], g2["post"]):
            assert np.max(np.abs(a["alpha"] - b["alpha"])) < 1e-5 * np.max(np.abs(b["alpha"]))
            assert np.max(np.abs(a["L"] - b["L"])) < 1e-8 * np.max(np.abs(b["L"]))


@pytest.mark.parametrize("N,S", [(1300, 2), (4500, 1)])
def test_variance_paths_and_nlz_beyond_the_wide_slab(va, N, S):
    D, K = 5, 6
    p = synth_problem(83, D, N, K, S)
    gp = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=4)
    vp = R.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    theta = np.concatenate([p["mu"].reshape(-1, order="F"), np.log(p["sigma"]), np.log(p["lam"]), p["eta"]])
    ref = R.negelcbo_vbmc(theta, 0, vp, gp, 0, False, 1, separate_K=True)
    got = va.negelcbo_batch(theta, 0, vp, gp, 0, False, 1, separate_K=True)
    assert abs(got["G"][0] - ref["G"]) < 1e-10 * max(1.0, abs(ref["G"]))
    assert abs(got["varG"][0] - ref["varG"]) < 1e-6 * max(abs(ref["varG"]), 1e-12) + 1e-9
    assert np.max(np.abs(got["J_sjk"][:, :, :, 0] - ref["J_sjk"])) < 1e-7 * max(1.0, np.max(np.abs(ref["J_sjk"])))
    refd = R.negelcbo_vbmc(theta, 0.5, vp, gp, 0, True, 2)
    gotd = va.negelcbo_batch(theta, 0.5, vp, gp, 0, True, 2)
    assert abs(gotd["F"][0] - refd["F"]) < 1e-8 * max(1.0, abs(refd["F"]))
    assert all(v < 1e-6 for v in block_relerr(gotd["dF"][:, 0], refd["dF"], D, K).values())
    # GP marginal likelihood + gradient (inv(K) as T'T with the narrow triangular inverse)
    gpd = {"X": p["X"], "y": p["y"], "s2": None, "covfun": 1, "Ncov": D + 1, "noisefun": (1, 0, 0), "Nnoise": 1, "meanfun": 4,
           "Nmean": 2 * D + 1, "meanfun_extras": None, "intmeanfun": 0}
    H = p["hyp"].copy()
    H[D + 1, :] = np.log(5e-2)
    nlz, dnlz = va.gplite_nlZ(H, gpd)
    nlz, dnlz = np.atleast_1d(nlz), np.asarray(dnlz).reshape(H.shape[0], -1)      # (S = 1: a scalar and a vector)
    for b in range(S):
        rz, rg = R.gplite_nlZ(H[:, b], gpd)
        assert abs(nlz[b] - rz) < 1e-9 * max(1.0, abs(rz))
        assert np.max(np.abs(dnlz[:, b] - rg)) < 1e-7 * max(1.0, np.max(np.abs(rg)))
