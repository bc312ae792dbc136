"""GPU parity: the fused acquisition sweep (acqwrapper_vbmc + acqf/acqflog/acqus/acqfsn2) vs the oracle."""
import numpy as np
import pytest

from oracle import vbmc_ref as R
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
        va.acqwrapper_vbmc(Xs, vp, gp, st, False, "acqviqr_vbmc", None)
    with pytest.raises(va.VbmcUnsupported):
        va.acqwrapper_vbmc(Xs, dict(vp, delta=np.array([0.1, 0.0, 0.0])), gp, st, False, "acqf_vbmc", None)
