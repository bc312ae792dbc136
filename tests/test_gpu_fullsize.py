"""GPU: BASELINE.json's full-size configurations (configs[2] / configs[4] shapes).

Full-size parity against the compiled C port of the reference loop nest (OpenMP), fed the exact device
RNG stream (vbmc_rng_dump), plus size-independent properties: bit-identical re-runs, value-only
kernel == value+gradient kernel, batch == singles."""
import numpy as np
import pytest

from oracle import c_oracle
from tests._cases import synth_problem
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
