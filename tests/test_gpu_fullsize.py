"""GPU: BASELINE.json's full-size configurations (configs[2] / configs[4] shapes).

Full-size parity against the compiled C port of the reference loop nest (OpenMP), fed the exact device
RNG stream (vbmc_rng_dump), plus size-independent properties: bit-identical re-runs, value-only
kernel == value+gradient kernel, batch == singles."""
import numpy as np
import pytest

from oracle import c_oracle
from oracle import vbmc_ref as R
from tests._cases import block_relerr, synth_problem
from tests.test_gpu_elbo import relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def va():
    import vbmc_amd

    return vbmc_amd


def setup(va, seed, D, N, K, S, noisy=False):
    p = synth_problem(seed, D, N, K, S, noisy=noisy)
    gp = va.gplite_post(p["hyp"], p["X"], p["y"], 1, 4, p["noisefun"], p["s2"])
    vp = va.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    theta = np.concatenate([p["mu"].reshape(-1, order="F"), np.log(p["sigma"]), np.log(p["lam"]), p["eta"]])
    return p, gp, vp, theta


@pytest.mark.parametrize("cfg", [("C3", 10, 400, 50, 20, 10000, False), ("C5", 20, 800, 100, 20, 20000, True)], ids=["C3", "C5"])
def test_full_size_parity_and_properties(va, cfg):
    name, D, N, K, S, Ns, noisy = cfg
    p, gp, vp, theta = setup(va, 41, D, N, K, S, noisy)
    eng = va.default_engine()
    seed = 2024
    a = va.negelcbo_batch(theta, 0, vp, gp, Ns, True, 0, seed=seed)
    b = va.negelcbo_batch(theta, 0, vp, gp, Ns, True, 0, seed=seed)
    assert np.array_equal(a["F"], b["F"]) and np.array_equal(a["dF"], b["dF"])          # run-to-run bit identical
    v = va.negelcbo_batch(theta, 0, vp, gp, Ns, False, 0, seed=seed)
    assert relerr(v["F"], a["F"]) < 1e-13 and relerr(v["H"], a["H"]) < 1e-13             # value-only kernel agrees
    # full-size parity: C port (OpenMP) on the dumped device stream
    eps = eng.ctx.rng_dump(D, K, 1, Ns, seed)[0]
    alpha = np.stack([q["alpha"] for q in gp["post"]], axis=1)
    Nnoise = 1
    F, dF, G, H = c_oracle.negelcbo(theta, p["X"], p["hyp"], alpha, eps, meanfun=4, Nnoise=Nnoise, openmp=True)
    assert relerr(a["G"][0], G) < 1e-10 and relerr(a["H"][0], H) < 1e-10 and relerr(a["F"][0], F) < 1e-10
    assert relerr(a["dF"][:, 0], dF) < 1e-9
    assert all(v < 1e-9 for v in block_relerr(a["dF"][:, 0], dF, D, K).values())   # each parameter group against its own scale
    # batch of jittered restarts == the same restarts one by one (shared seed -> restart r uses stream r)
    th = theta[:, None] + 0.02 * np.random.default_rng(1).standard_normal((theta.size, 3))
    bt = va.negelcbo_batch(th, 0, vp, gp, Ns, True, 0, seed=seed)
    assert np.all(np.isfinite(bt["F"])) and np.all(np.isfinite(bt["dF"]))
    assert relerr(bt["G"][1], va.negelcbo_batch(th[:, 1], 0, vp, gp, 0, False, 0)["G"][0]) < 1e-13


def test_c5_shape_small_sample_against_numpy_oracle(va):
    """D=20, K=100 (KT=7, QS=6 MFMA instantiation) with provided-noise GP, against the NumPy oracle."""
    from oracle import vbmc_ref as R

    p = synth_problem(43, 20, 120, 100, 2, noisy=True)
    gp = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=4, noisefun=p["noisefun"], s2=p["s2"])
    vp = R.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    theta = np.concatenate([p["mu"].reshape(-1, order="F"), np.log(p["sigma"]), np.log(p["lam"]), p["eta"]])
    eps = np.random.default_rng(2).standard_normal((100, 25, 20))
    ref = R.negelcbo_vbmc(theta, 0, vp, gp, 50, True, 0, eps=eps)
    F, dF = va.negelcbo_vbmc(theta, 0, vp, gp, 50, 1, 0, eps=eps)
    assert relerr(F, ref["F"]) < 1e-10 and relerr(dF, ref["dF"]) < 1e-9


def test_c5_noisy_path_prediction_and_iqr_acquisition(va):
    """BASELINE configs[4] GP shape (D=20, N=800, noisy likelihood with user-supplied s2): gplite_post on the device,
    gplite_pred and the VIQR / IMIQR acquisition functions against the oracle, fp64 end to end."""
    p = synth_problem(5, 20, 800, 100, 3, noisy=True)
    gp_o = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=p["meanfun"], noisefun=p["noisefun"], s2=p["s2"])
    gp = va.gplite_post(p["hyp"], p["X"], p["y"], 1, p["meanfun"], p["noisefun"], p["s2"])
    for a, b in zip(gp["post"], gp_o["post"]):
        assert relerr(a["alpha"], b["alpha"]) < 1e-8 and a["Lchol"] == b["Lchol"] and a["sn2_mult"] == b["sn2_mult"]
    rng = np.random.default_rng(1)
    D = 20
    Xs = 1.3 * rng.standard_normal((150, D))
    Xa = 1.3 * rng.standard_normal((100, D))
    r_o = R.gplite_pred(gp_o, Xs, None, None, True)
    r_d = va.gplite_pred(gp, Xs, None, None, True)
    sf2 = np.exp(2 * gp_o["post"][0]["hyp"][D])
    assert relerr(r_d[2], r_o[2]) < 1e-8 and np.max(np.abs(np.asarray(r_d[3]) - np.asarray(r_o[3]))) < 1e-9 * sf2
    gl = np.exp(np.mean(np.stack([q["hyp"][:D] for q in gp_o["post"]], axis=1), axis=1))
    extra = dict(X_rescaled=p["X"] / gl[None, :], sn2new=np.asarray(p["s2"]) + 0.01)
    gp_o = dict(gp_o, **extra)
    gp = dict(gp, **extra)
    vp = R.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    Kax, Ct = R.acq_is_precompute(gp_o, Xa)
    fs2a = np.asarray(R.gplite_pred(gp_o, Xa, None, None, True)[3]).reshape(100, -1)
    lnw = 0.5 * rng.standard_normal((3, 100))
    st = {"ymax": float(np.max(p["y"])), "VarianceRegularizedAcqFcn": False, "TolGPVar": 1e-4, "gplengthscale": gl}
    ais_o = {"Xa": Xa, "Kax_mat": Kax, "Ctmp_mat": Ct, "fs2a": fs2a, "lnw": lnw}
    for name in ("acqviqr", "acqimiqr"):
        ref, _, _ = R.acqwrapper_vbmc(Xs, vp, gp_o, dict(st, ActiveImportanceSampling=ais_o), name)
        ais_d = {"Xa": Xa, "lnw": (np.zeros_like(lnw) if name == "acqviqr" else lnw)}      # Ctmp / fs2a built on the device
        acq = va.acqwrapper_vbmc(Xs, vp, gp, dict(st, ActiveImportanceSampling=ais_d), False, name + "_vbmc", None)
        assert np.max(np.abs(acq - ref)) < 1e-7, (name, np.max(np.abs(acq - ref)))
