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


@pytest.mark.parametrize("cfg", [(32, 40, 160, 2, 24), (32, 40, 150, 1, 0), (24, 30, 250, 1, 0), (28, 30, 190, 1, 12)])
def test_mixtures_whose_finalize_record_exceeds_the_lds(va, cfg):
    D, N, K, S, Ns = cfg
    assert 4 * D * K + 9 * K > 19400
    gp, vp, theta, tb = _problem(72, D, N, K, S)
    eps = np.random.default_rng(4).standard_normal((K, max(Ns, 2) // 2, D)) if Ns else None
    ref = R.negelcbo_vbmc(theta, 0, vp, gp, Ns, True, 0, thetabnd=tb, eps=eps)
    got = va.negelcbo_batch(theta, 0, vp, gp, Ns, True, 0, tb, eps=eps)
    assert abs(got["F"][0] - ref["F"]) < 1e-10 * max(1.0, abs(ref["F"]))
    assert abs(got["G"][0] - ref["G"]) < 1e-10 * max(1.0, abs(ref["G"])) and abs(got["H"][0] - ref["H"]) < 1e-10 * max(1.0, abs(ref["H"]))
    for key in ("dF", "dG", "dH"):
        assert all(v < 1e-9 for v in block_relerr(got[key][:, 0], ref[key], D, K).values()), key
