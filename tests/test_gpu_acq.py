"""GPU parity: the fused acquisition sweep (acqwrapper_vbmc + acqf/acqflog/acqus/acqfsn2) vs the oracle."""
import numpy as np
import pytest

from oracle import vbmc_ref as R
from tests._cases import acq_golden_cases, load_acq_golden
from tests.test_gpu_elbo import problem, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def va():
    import vbmc_amd

    return vbmc_amd


def setup(seed, D, N, K, S):
    p, gp, vp, _ = problem(seed, D, N, K, S)
    rng = np.random.default_rng(seed + 100)
    Xs = np.vstack([1.2 * rng.standard_normal((300, D)), gp["X"][:20] + 1e-3 * rng.standard_normal((20, D)), gp["X"][:5]])
    st = {"ymax": float(np.max(gp["y"])), "VarianceRegularizedAcqFcn": True, "TolGPVar": 1e-4}
    return gp, vp, Xs, st, rng


@pytest.mark.parametrize("name", ["acqf", "acqflog", "acqus"])
@pytest.mark.parametrize("shape", [(4, 60, 5, 3), (10, 200, 12, 4), (2, 30, 3, 1)])
def test_acq_matches_oracle(va, name, shape):
    gp, vp, Xs, st, rng = setup(7, *shape)
    outside = rng.random(Xs.shape[0]) < 0.1
    ref, fbar_r, vtot_r = R.acqwrapper_vbmc(Xs, vp, gp, st, name, outside)
    acq, fbar, vtot = va.acqwrapper_vbmc(Xs, vp, gp, st, False, name + "_vbmc", None, outside=outside, nargout=3)
    assert np.array_equal(np.isinf(acq), outside)
    ok = ~outside
    assert relerr(fbar, fbar_r) < 1e-10
    # near training inputs fs2 = kss - |V|^2 cancels to ~1e-10 of kss: compare vtot on the scale of the prior variance
    sf2 = np.exp(2 * gp["post"][0]["hyp"][shape[0]])
    assert np.max(np.abs(vtot - vtot_r)) < 1e-9 * sf2
    # fs2's cancellation error dv is amplified by the regulariser: d(acq)/acq ~ (TolVar/vtot^2 + 1/vtot) dv
    dv = 1e-9 * sf2
    sel = ok & (vtot_r > 1e-7 * sf2)
    assert sel.sum() > 200
    amp = (st["TolGPVar"] * (vtot_r < st["TolGPVar"]) / vtot_r**2 + 1.0 / vtot_r) * dv
    if name == "acqflog":
        assert np.all(np.abs(acq[sel] - ref[sel]) <= 1e-9 * (1 + np.abs(ref[sel])) + amp[sel])
    else:
        assert np.all(np.abs(acq[sel] - ref[sel]) <= (1e-9 + amp[sel]) * np.abs(ref[sel]) + 1e-300)
    assert np.any(vtot_r[ok] < st["TolGPVar"])        # the regularised branch (:35-45) is exercised
    # transposed call form used by CMA-ES (:5,54)
    accT = va.acqwrapper_vbmc(Xs[:7].T, vp, gp, st, True, name + "_vbmc", None)
    assert accT.shape == (1, 7) and np.allclose(accT.reshape(-1), acq[:7], rtol=1e-12, atol=0) | np.isinf(acq[:7]).all()


def test_acqfsn2_nearest_neighbour_noise(va):
    gp, vp, Xs, st, rng = setup(9, 5, 80, 6, 3)
    gl = np.exp(np.mean(np.stack([p["hyp"][:5] for p in gp["post"]], axis=1), axis=1))
    gp = dict(gp, X_rescaled=gp["X"] / gl[None, :], sn2new=0.01 + 0.1 * rng.random(80))
    st = dict(st, gplengthscale=gl, VarianceRegularizedAcqFcn=False)
    Xs = Xs[:300]                                     # generic points: no exact nearest-neighbour ties
    ref, _, vtot_r = R.acqwrapper_vbmc(Xs, vp, gp, st, "acqfsn2")
    acq = va.acqwrapper_vbmc(Xs, vp, gp, st, False, "acqfsn2_vbmc", None)
    assert relerr(acq, ref) < 1e-8


def test_acq_unsupported_forms(va):
    gp, vp, Xs, st, _ = setup(3, 3, 20, 2, 2)
    with pytest.raises(va.VbmcUnsupported):
        va.acqwrapper_vbmc(Xs, vp, gp, st, False, "acqeig_vbmc", None)
    with pytest.raises(va.VbmcUnsupported):
        va.acqwrapper_vbmc(Xs, dict(vp, delta=np.array([0.1, 0.0, 0.0])), gp, st, False, "acqf_vbmc", None)


def iqr_setup(va, seed, D, N, K, S, Na, per_s=False, lnw_zero=True):
    gp, vp, Xs, st, rng = setup(seed, D, N, K, S)
    gl = np.exp(np.mean(np.stack([p["hyp"][:D] for p in gp["post"]], axis=1), axis=1))
    gp = dict(gp, X_rescaled=gp["X"] / gl[None, :], sn2new=0.02 + 0.1 * rng.random(N))
    Xa = 1.1 * rng.standard_normal((Na, D, S)) if per_s else 1.1 * rng.standard_normal((Na, D))
    Kax, Ct = R.acq_is_precompute(gp, Xa)
    if per_s:
        fs2a = np.stack([np.asarray(R.gplite_pred(gp, Xa[:, :, s], None, None, True)[3]).reshape(Na, -1)[:, s] for s in range(S)], axis=1)
    else:
        fs2a = np.asarray(R.gplite_pred(gp, Xa, None, None, True)[3]).reshape(Na, -1)
    lnw = np.zeros((S, Na)) if lnw_zero else 0.7 * rng.standard_normal((S, Na))
    ais = {"Xa": Xa, "Kax_mat": Kax, "Ctmp_mat": Ct, "fs2a": fs2a, "lnw": lnw}
    st = dict(st, gplengthscale=gl, ActiveImportanceSampling=ais)
    return gp, vp, Xs, st


@pytest.mark.parametrize("cfg", [(4, 60, 5, 3, 30), (10, 200, 12, 4, 100), (3, 40, 3, 1, 16), (6, 90, 4, 2, 37)])
def test_acqviqr_matches_oracle(va, cfg):
    D, N, K, S, Na = cfg
    gp, vp, Xs, st = iqr_setup(va, 11, D, N, K, S, Na)
    ref, fbar_r, vtot_r = R.acqwrapper_vbmc(Xs, vp, gp, st, "acqviqr")
    # state computed entirely on the device (Ctmp, fs2a from Xa) ...
    st_dev = dict(st, ActiveImportanceSampling={"Xa": st["ActiveImportanceSampling"]["Xa"]})
    acq, fbar, vtot = va.acqwrapper_vbmc(Xs, vp, gp, st_dev, False, "acqviqr_vbmc", None, nargout=3)
    # ... and uploaded as the reference's optimState.ActiveImportanceSampling holds it
    acq2 = va.acqwrapper_vbmc(Xs, vp, gp, st, False, "acqviqr_vbmc", None)
    far = vtot_r > st["TolGPVar"]             # away from the regulariser's amplification (see test_acq_matches_oracle)
    assert far.sum() > 200
    assert np.max(np.abs(acq[far] - ref[far])) < 1e-8 and np.max(np.abs(acq2[far] - ref[far])) < 1e-8
    assert relerr(fbar, fbar_r) < 1e-10
    near = ~far
    assert np.all(np.isfinite(acq[near])) and np.all(acq[near] >= ref[near] - 1e-3 * np.abs(ref[near]) - 1.0)


@pytest.mark.parametrize("per_s", [False, True])
def test_acqimiqr_matches_oracle(va, per_s):
    gp, vp, Xs, st = iqr_setup(va, 13, 5, 80, 6, 3, 50, per_s=per_s, lnw_zero=False)
    ref, _, vtot_r = R.acqwrapper_vbmc(Xs, vp, gp, st, "acqimiqr")
    ais = dict(st["ActiveImportanceSampling"])
    ais.pop("Ctmp_mat")                        # IMIQR does not cache Ctmp (acqimiqr_vbmc.m:77): computed on the device
    acq = va.acqwrapper_vbmc(Xs, vp, gp, dict(st, ActiveImportanceSampling=ais), False, "acqimiqr_vbmc", None)
    far = vtot_r > st["TolGPVar"]
    assert np.max(np.abs(acq[far] - ref[far])) < 1e-8
    info = va.acq_info("acqimiqr_vbmc")
    assert info["log_flag"] and info["importance_sampling"] and not info["variational_importance_sampling"]


def test_noisy_gp_prediction_and_viqr_without_s2star(va):
    """Noisy targets (noisefun [1 1 0], user-supplied s2 at the training points): the acquisition sweep calls
    gplite_pred with empty s2star, which counts as zero (gplite_noisefun.m:51); VIQR on such a GP."""
    p, gp, vp, _ = problem(21, 5, 70, 4, 3, noisy=True)
    assert tuple(gp["noisefun"])[:2] == (1, 1) and gp["s2"] is not None
    rng = np.random.default_rng(5)
    Xs = 1.2 * rng.standard_normal((200, 5))
    ymu, ys2, fmu, fs2 = va.gplite_pred(gp, Xs, None, None, True)
    r = R.gplite_pred(gp, Xs, None, None, True)
    sf2 = np.exp(2 * gp["post"][0]["hyp"][5])
    assert relerr(fmu, r[2]) < 1e-9 and np.max(np.abs(np.asarray(ys2) - np.asarray(r[1]))) < 1e-9 * sf2
    gl = np.exp(np.mean(np.stack([q["hyp"][:5] for q in gp["post"]], axis=1), axis=1))
    gp = dict(gp, X_rescaled=gp["X"] / gl[None, :], sn2new=np.asarray(gp["s2"]) + 0.01)
    Xa = 1.1 * rng.standard_normal((40, 5))
    st = {"ymax": float(np.max(gp["y"])), "VarianceRegularizedAcqFcn": False, "TolGPVar": 1e-4, "gplengthscale": gl,
          "ActiveImportanceSampling": {"Xa": Xa}}
    Kax, Ct = R.acq_is_precompute(gp, Xa)
    fs2a = np.asarray(R.gplite_pred(gp, Xa, None, None, True)[3]).reshape(40, -1)
    st_ref = dict(st, ActiveImportanceSampling={"Xa": Xa, "Ctmp_mat": Ct, "fs2a": fs2a, "lnw": np.zeros((3, 40))})
    ref, _, _ = R.acqwrapper_vbmc(Xs, vp, gp, st_ref, "acqviqr")
    acq = va.acqwrapper_vbmc(Xs, vp, gp, st, False, "acqviqr_vbmc", None)
    assert np.max(np.abs(acq - ref)) < 1e-8


def test_viqr_end_to_end_through_the_mirror(va):
    """activeimportancesampling_vbmc (draws from the variational posterior) -> acqwrapper_vbmc(acqviqr): everything
    but the draws on the device; the oracle gets the same importance points."""
    gp, vp, Xs, st, rng = setup(17, 4, 70, 5, 3)
    gl = np.exp(np.mean(np.stack([q["hyp"][:4] for q in gp["post"]], axis=1), axis=1))
    gp = dict(gp, X_rescaled=gp["X"] / gl[None, :], sn2new=np.full(70, 0.05))
    ais = va.activeimportancesampling_vbmc(vp, gp, "acqviqr_vbmc", None, {"ActiveImportanceSamplingMCMCSamples": 64},
                                           rng=np.random.default_rng(3))
    assert ais["Xa"].shape == (64, 4) and ais["lnw"].shape == (3, 64)
    st = dict(st, gplengthscale=gl, VarianceRegularizedAcqFcn=False, ActiveImportanceSampling=ais)
    acq = va.acqwrapper_vbmc(Xs, vp, gp, st, False, "acqviqr_vbmc", None)
    Kax, Ct = R.acq_is_precompute(gp, ais["Xa"])
    fs2a = np.asarray(R.gplite_pred(gp, ais["Xa"], None, None, True)[3]).reshape(64, -1)
    st_o = dict(st, ActiveImportanceSampling={"Xa": ais["Xa"], "Ctmp_mat": Ct, "fs2a": fs2a, "lnw": ais["lnw"]})
    ref, _, _ = R.acqwrapper_vbmc(Xs, vp, gp, st_o, "acqviqr")
    assert np.max(np.abs(acq - ref)) < 1e-8
    assert int(np.argmin(acq)) == int(np.argmin(ref))


@pytest.mark.parametrize("path", acq_golden_cases())
def test_acquisition_golden_vectors(va, path):
    """Device acquisition sweep against the committed 50-digit mpmath vectors: prediction, vbmc_pdf, the four closed-form
    acquisition functions and VIQR with Ctmp / fs2a computed on the device from the importance points alone."""
    vp, gp, Xs, st, exp = load_acq_golden(path)
    for name in ("acqf", "acqflog", "acqus", "acqfsn2", "acqviqr"):
        acq, fbar, vtot = va.acqwrapper_vbmc(Xs, vp, gp, st, False, name + "_vbmc", None, nargout=3)
        tol = 1e-8 if name == "acqviqr" else 1e-9
        assert np.max(np.abs(acq - exp[name]) / np.maximum(1e-300, np.abs(exp[name]))) < tol, (name, acq, exp[name])
        assert np.max(np.abs(fbar - exp["fbar"])) < 1e-10 and np.max(np.abs(vtot - exp["vtot"]) / exp["vtot"]) < 1e-9


def test_proposal_weights_match_the_oracle(va):
    """activesample_proposalpdf (private/activeimportancesampling_vbmc.m:301-340): log weights of points drawn from the smoothed
    variational posterior and from the box-uniforms, GP prediction on the device, against the oracle's point-by-point form --
    both acquisition functions, with and without the variational density in the base, pure-VP / pure-box / mixed proposals."""
    from vbmc_amd.acq import _proposal_lnw

    gp, vp, Xs, st, rng = setup(23, 3, 40, 4, 3)
    D, K, S_ = 3, 4, 3
    X = gp["X"]
    rect = 2 * np.std(X, axis=0, ddof=1)
    sc = (0.05, 0.2, 1.0)
    vp_is = dict(vp, K=4 * K, w=np.tile(vp["w"], 4) / 4, mu=np.tile(vp["mu"], (1, 4)),
                 sigma=np.concatenate([vp["sigma"]] + [np.sqrt(vp["sigma"] ** 2 + c * c) for c in sc]))
    Xa = np.concatenate([X[rng.integers(0, 40, 12)] + (2 * rng.random((12, D)) - 1) * rect, 1.2 * rng.standard_normal((10, D)),
                         50.0 + rng.standard_normal((2, D))], axis=0)      # the last two: outside every box
    eng = va.default_engine()
    for name, oname in (("acqimiqr_vbmc", "acqimiqr"), ("acqviqr_vbmc", "acqviqr")):
        for w_vp in (0.5, 1.0, 0.0):
            for isvp in (False, True):
                lw, f2 = _proposal_lnw(Xa, gp, vp_is, w_vp, rect, name, vp, isvp, eng)
                lo, f2o = R.activesample_proposalpdf(Xa, gp, vp_is, w_vp, rect, oname, vp, isvp)
                assert relerr(f2, f2o) < 1e-9
                fin = np.isfinite(lo)            # (a point outside every box and far from the mixture has density 0: NaN / Inf weights on
                assert np.array_equal(fin, np.isfinite(lw))      # both sides, which :146 then turns into -Inf)
                assert fin.sum() >= 22 * S_ and np.max(np.abs(lw[fin] - lo[fin])) < 1e-8 * max(1.0, np.max(np.abs(lo[fin])))


def test_imiqr_end_to_end_through_the_mirror(va):
    """activeimportancesampling_vbmc for acqimiqr_vbmc (round 3: importance sampling-resampling + MCMC per GP hyper-sample with every
    log-density evaluation a batched device prediction) -> acqwrapper_vbmc(acqimiqr).  The sampler is a stand-in for
    utils/eissample_lite.m, so what is checked is what the reference's OUTPUT must satisfy: shapes (Xa Na x D x S, lnw S x Na),
    every point inside the sampler's box, lnw = islogf1 - log p of the chain's own target (recomputed with the ORACLE's prediction),
    chains that sit where their target has mass (mean log density of the samples far above that of the box's uniform points), and the
    acquisition sweep on that state equal to the oracle's on the same state."""
    D, N, K, S = 3, 60, 4, 3
    gp, vp, Xs, st, rng = setup(29, D, N, K, S)
    gl = np.exp(np.mean(np.stack([q["hyp"][:D] for q in gp["post"]], axis=1), axis=1))
    gp = dict(gp, X_rescaled=gp["X"] / gl[None, :], sn2new=np.full(N, 0.05))
    opts = {"ActiveImportanceSamplingMCMCSamples": 48, "ActiveImportanceSamplingVPSamples": 40, "ActiveImportanceSamplingBoxSamples": 40}
    ais = va.activeimportancesampling_vbmc(vp, gp, "acqimiqr_vbmc", None, opts, rng=np.random.default_rng(4))
    Xa, lnw = ais["Xa"], ais["lnw"]
    assert Xa.shape == (48, D, S) and lnw.shape == (S, 48) and ais["fs2a"].shape == (48, S)
    X = gp["X"]
    diam = X.max(axis=0) - X.min(axis=0)
    assert np.all(Xa >= (X.min(axis=0) - 0.5 * diam)[None, :, None]) and np.all(Xa <= (X.max(axis=0) + 0.5 * diam)[None, :, None])
    u = 0.6745
    uni = (X.min(axis=0) - 0.5 * diam) + rng.random((400, D)) * 2 * diam
    for s in range(S):
        pr = R.gplite_pred(gp, Xa[:, :, s], None, None, True)
        fmu = np.asarray(pr[2]).reshape(48, S)[:, s]
        # the chain's target (log_isbasefun, private/activeimportancesampling_vbmc.m:343-353) reads the FIRST two outputs of gplite_pred:
        # ymu and ys2, the predictive variance with the observation noise; the weight's numerator islogf1 the latent mean (:209-221)
        ym = np.asarray(pr[0]).reshape(48, S)[:, s]
        fs = np.sqrt(np.asarray(pr[1]).reshape(48, S)[:, s])
        logp = ym + u * fs + np.log1p(-np.exp(-2 * u * fs))             # acq/acqimiqr_vbmc.m:24-27
        assert np.max(np.abs(lnw[s] - (fmu - logp))) < 1e-7 * max(1.0, np.max(np.abs(logp)))
        pu = R.gplite_pred(gp, uni, None, None, True)
        fu = np.asarray(pu[0]).reshape(400, S)[:, s]
        su = np.sqrt(np.asarray(pu[1]).reshape(400, S)[:, s])
        assert np.mean(logp) > np.mean(fu + u * su + np.log1p(-np.exp(-2 * u * su))) + 1.0
    st = dict(st, gplengthscale=gl, VarianceRegularizedAcqFcn=False, ActiveImportanceSampling=ais)
    acq = va.acqwrapper_vbmc(Xs, vp, gp, st, False, "acqimiqr_vbmc", None)
    Kax, Ct = R.acq_is_precompute(gp, Xa)
    st_o = dict(st, ActiveImportanceSampling={"Xa": Xa, "Kax_mat": Kax, "Ctmp_mat": Ct, "fs2a": ais["fs2a"], "lnw": lnw})
    ref, _, _ = R.acqwrapper_vbmc(Xs, vp, gp, st_o, "acqimiqr")
    assert np.max(np.abs(acq - ref)) < 1e-8 * max(1.0, np.max(np.abs(ref)))
    assert int(np.argmin(acq)) == int(np.argmin(ref))
    assert ais["funccount"] > 0
