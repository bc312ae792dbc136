"""GPU parity: gplite_nlZ (+ gradient) through the C ABI vs the mpmath golden vectors and the oracle."""
import numpy as np
import pytest

from oracle import vbmc_ref as R
from tests._cases import load_nlz_golden, nlz_golden_cases
from tests.test_gpu_elbo import relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def va():
    import vbmc_amd

    return vbmc_amd


def make_gp(rng, N, D, meanfun, noisefun, low_noise=False):
    X = 1.5 * rng.standard_normal((N, D))
    y = -0.5 * np.sum((X / 1.3) ** 2, axis=1) + 0.3 * np.sin(X[:, 0]) + 0.05 * rng.standard_normal(N)
    s2 = 0.01 + 0.05 * rng.random(N) if noisefun[1] else None
    gp = {"X": X, "y": y, "s2": s2, "covfun": 1, "Ncov": D + 1, "noisefun": tuple(noisefun), "Nnoise": R.noisefun_nhyp(noisefun),
          "meanfun": meanfun, "Nmean": R.meanfun_nhyp(meanfun, D), "meanfun_extras": None, "intmeanfun": 0}
    Nhyp = gp["Ncov"] + gp["Nnoise"] + gp["Nmean"]

    def draw():
        h = np.zeros(Nhyp)
        h[:D] = np.log(0.8) + 0.3 * rng.standard_normal(D)
        h[D] = np.log(np.std(y)) + 0.2 * rng.standard_normal()
        h[D + 1] = (np.log(3e-4) if low_noise else np.log(5e-2)) + 0.2 * rng.standard_normal()
        i = D + 2
        if noisefun[1] == 2:
            h[i] = 0.3 * rng.standard_normal()
            i += 1
        if noisefun[2] == 1:
            h[i] = np.median(y)
            h[i + 1] = np.log(0.1)
            i += 2
        if meanfun >= 1:
            h[i] = np.max(y) + 0.1 * rng.standard_normal()
        if meanfun == 4:
            h[i + 1 : i + 1 + D] = 0.2 * rng.standard_normal(D)
            h[i + 1 + D :] = np.log(2.0) + 0.1 * rng.standard_normal(D)
        return h

    return gp, draw


@pytest.mark.parametrize("path", nlz_golden_cases())
def test_nlz_golden(va, path):
    gp, hyp, exp = load_nlz_golden(path)
    nlZ, dnlZ = va.gplite_nlZ(hyp, gp) if hyp.shape[1] > 1 else va.gplite_nlZ(hyp[:, 0], gp)
    assert relerr(np.atleast_1d(nlZ), exp["nlZ"]) < 1e-11
    assert relerr(np.asarray(dnlZ).reshape(hyp.shape[0], -1), exp["dnlZ"].T) < 1e-9
    v = va.gplite_nlZ(hyp[:, 0], gp, nargout=1)            # value-only form
    assert abs(v - exp["nlZ"][0]) < 1e-11 * max(1.0, abs(exp["nlZ"][0]))


@pytest.mark.parametrize("cfg", [(60, 4, 4, (1, 0, 0)), (130, 7, 4, (1, 1, 0)), (47, 3, 1, (1, 2, 0)), (90, 5, 0, (1, 0, 1)),
                                 (400, 10, 4, (1, 0, 0)), (33, 2, 4, (1, 2, 1))])
def test_nlz_batch_matches_oracle(va, cfg):
    N, D, meanfun, noisefun = cfg
    rng = np.random.default_rng(N + D)
    gp, draw = make_gp(rng, N, D, meanfun, noisefun)
    B = 5
    H = np.stack([draw() for _ in range(B)], axis=1)
    Nhyp = H.shape[0]
    hp = {"mu": np.zeros(Nhyp), "sigma": 3.0 * np.ones(Nhyp), "df": np.array(([3.0, 0.0, np.inf] * Nhyp)[:Nhyp])}
    nlZ, dnlZ = va.gplite_nlZ(H, gp, hp)
    for b in range(B):
        f, g = R.gplite_nlZ(H[:, b], gp, hp)
        assert abs(nlZ[b] - f) < 1e-10 * max(1.0, abs(f)), (b, nlZ[b], f)
        assert relerr(dnlZ[:, b], g) < 1e-8, (b, relerr(dnlZ[:, b], g))
    # the single-vector reference form
    f1, g1 = va.gplite_nlZ(H[:, 2], gp, hp)
    assert f1 == nlZ[2] and np.array_equal(g1, dnlZ[:, 2])


@pytest.mark.parametrize("N,D", [(16, 2), (17, 3), (48, 13), (81, 20), (70, 32), (1080, 6), (520, 5), (1001, 4), (1024, 3), (1030, 3)])
def test_nlz_inverse_kernel_shapes(va, N, D):
    """Kinv = L\\(L'\\eye(N)) (gplite_core.m:240) is formed as T'T, T = inv(L') (k_tri_inverse + k_syrk_tt on 64 x 64 tiles):
    a single 16-block (N <= 16), N not a multiple of the tile sizes, D padded to the next kernel instantiation
    (13 -> 16, 20 -> 24, 32), and N = 1080 near the LDS limit of the triangular-solve slab.  Round 5: the workgroup-per-slab
    inverse (k_tri_inverse2) keeps 4 or 8 row blocks per wave in registers -- N = 520 and 1001 / 1024 take the 8-slot
    instantiation (the last one at its limit of 64 blocks), N = 1030 falls back to the one-wave kernel."""
    rng = np.random.default_rng(N)
    gp, draw = make_gp(rng, N, D, 4, (1, 0, 0))
    H = np.stack([draw() for _ in range(2)], axis=1)
    nlZ, dnlZ = va.gplite_nlZ(H, gp)
    for b in range(2):
        f, g = R.gplite_nlZ(H[:, b], gp)
        assert abs(nlZ[b] - f) < 1e-10 * max(1.0, abs(f))
        assert relerr(dnlZ[:, b], g) < 1e-8


def test_nlz_low_noise_branch_and_retries(va):
    """min(sn2) < 1e-6 takes the Lchol = false branch (gplite_core.m:84-99); duplicated inputs with tiny
    noise force the x10 noise-inflation retries (:91-94), and sn2_mult enters the noise gradient (:257-262)."""
    rng = np.random.default_rng(5)
    gp, draw = make_gp(rng, 40, 3, 4, (1, 0, 0), low_noise=True)
    H = np.stack([draw() for _ in range(3)], axis=1)
    nlZ, dnlZ = va.gplite_nlZ(H, gp)
    for b in range(3):
        f, g = R.gplite_nlZ(H[:, b], gp)
        assert abs(nlZ[b] - f) < 1e-9 * max(1.0, abs(f)) and relerr(dnlZ[:, b], g) < 1e-7
    # exact duplicates + sn2 = 1e-18: K + sn2*I is singular to rounding, so chol fails until the noise has been
    # inflated enough (:91-94).  How many x10 steps that takes is decided by rounding in the factorisation (it is
    # in MATLAB, too), and the accepted matrix has condition ~1e16/mult -- only coarse agreement is meaningful.
    gp2 = dict(gp)
    gp2["X"] = np.vstack([gp["X"], gp["X"][:6]])
    gp2["y"] = np.concatenate([gp["y"], gp["y"][:6]])
    h = draw()
    h[3 + 1] = np.log(1e-9)
    post = va.gplite_post(h, gp2["X"], gp2["y"], 1, 4, (1, 0, 0))["post"][0]
    assert post["sn2_mult"] > 1.0 and not post["Lchol"]
    f, g = R.gplite_nlZ(h, gp2)
    fh, gh = va.gplite_nlZ(h, gp2)
    assert np.isfinite(fh) and np.all(np.isfinite(gh)) and abs(fh - f) < 0.1 * abs(f)


def test_nlz_gradient_is_the_derivative(va):
    """Central differences of the device value against the device gradient (smooth noise models)."""
    rng = np.random.default_rng(9)
    gp, draw = make_gp(rng, 70, 4, 4, (1, 1, 0))
    h = draw()
    f, g = va.gplite_nlZ(h, gp)
    e = 1e-5
    Hp = np.stack([h + e * np.eye(h.size)[i] for i in range(h.size)] + [h - e * np.eye(h.size)[i] for i in range(h.size)], axis=1)
    v = va.gplite_nlZ(Hp, gp, nargout=1)
    fd = (v[: h.size] - v[h.size :]) / (2 * e)
    assert relerr(fd, g) < 1e-6


def test_nlz_errors(va):
    rng = np.random.default_rng(1)
    gp, draw = make_gp(rng, 20, 2, 4, (1, 0, 0))
    with pytest.raises(ValueError, match="gplite_nlZ:dimmismatch"):
        va.gplite_nlZ(np.zeros(3), gp)
    bad = dict(gp, meanfun=6, Nmean=gp["Nmean"])
    with pytest.raises(va.VbmcUnsupported):
        va.gplite_nlZ(draw(), bad)
