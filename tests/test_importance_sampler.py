"""The host side of the importance sampler behind acqimiqr_vbmc (private/activeimportancesampling_vbmc.m:103-246): the ensemble
slice sampler that stands where the reference calls utils/eissample_lite.m.  Pure host logic -- the log density is a NumPy
function here -- so the checks run on every CPU test pass; the device end-to-end checks are in tests/test_gpu_acq.py."""
import numpy as np

from vbmc_amd.acq import _islogf, _vbmc_lnpdf, ensemble_slice_sample
from oracle import vbmc_ref as R


import pytest


@pytest.mark.parametrize("spec", [3, 1])
def test_ensemble_slice_sampler_recovers_correlated_gaussians(spec):
    """Three targets at once (one ensemble each), each a correlated 3-D Gaussian with its own mean / covariance, inside a wide box:
    first and second moments of 3000 recorded samples per target (thin 2) within Monte-Carlo error of the truth, the recorded
    log densities are the target's own, every sample inside the bounds -- with the stepping-out steps / shrinkage proposals evaluated
    three per batched call (the default) and one at a time (the textbook procedure)."""
    rng = np.random.default_rng(5)
    D, E, W = 3, 3, 8
    means = rng.standard_normal((E, D))
    Ls = [np.linalg.cholesky(np.array([[1.0, 0.6, 0.2], [0.6, 1.5, -0.3], [0.2, -0.3, 0.7]]) * (0.5 + e)) for e in range(E)]
    P = [np.linalg.inv(L @ L.T) for L in Ls]

    def logp(X, e):
        d = X - means[e]
        return np.array([-0.5 * d[i] @ P[e[i]] @ d[i] for i in range(X.shape[0])])

    x0 = means[:, None, :] + 0.5 * rng.standard_normal((E, W, D))
    N = 3000
    Xs, lps = ensemble_slice_sample(logp, x0, N, -20 * np.ones(D), 20 * np.ones(D), thin=2, burnin=400, rng=rng, spec=spec)
    assert Xs.shape == (E, N, D) and lps.shape == (E, N)
    assert np.all(np.abs(Xs) <= 20)
    for e in range(E):
        C = Ls[e] @ Ls[e].T
        assert np.allclose(lps[e], logp(Xs[e], np.full(N, e)), rtol=0, atol=1e-12)
        se = np.sqrt(np.diag(C) / (N / 12.0))          # autocorrelated chain: allow an effective sample size of N / 12
        assert np.all(np.abs(np.mean(Xs[e], axis=0) - means[e]) < 4 * se), (e, np.mean(Xs[e], axis=0), means[e])
        Chat = np.cov(Xs[e].T)
        assert np.max(np.abs(Chat - C)) < 0.3 * np.max(np.abs(C)), (e, Chat, C)


def test_ensemble_slice_sampler_respects_hard_bounds_and_counts_moves():
    """A density that is flat inside the box [0, 1]^2 and a box that cuts it: every sample inside; burn-in and thinning counted per
    WALKER MOVE as utils/eissample_lite.m does (:386 one recorded point per iteration): N samples need burnin + N thin moves."""
    rng = np.random.default_rng(1)
    calls = {"n": 0}

    def logp(X, e):
        calls["n"] += X.shape[0]
        return np.zeros(X.shape[0])

    x0 = 0.25 + 0.5 * rng.random((2, 6, 2))
    Xs, lps, info = ensemble_slice_sample(logp, x0, 50, np.zeros(2), np.ones(2), thin=3, burnin=20, rng=rng, return_info=True)
    assert Xs.shape == (2, 50, 2) and np.all((Xs >= 0) & (Xs <= 1)) and np.all(lps == 0)
    assert np.std(Xs[0][:, 0]) > 0.15                      # it moves: a uniform on [0, 1] has 0.29
    assert info["funccount"] == calls["n"]   # out-of-bounds proposals are rejected without an evaluation


def test_islogf_and_lnpdf_match_the_oracle():
    """acqfun('islogf*') of both importance-sampled acquisition functions and log vbmc_pdf against the oracle's point-by-point forms."""
    rng = np.random.default_rng(2)
    Na, S, D, K = 7, 3, 4, 5
    fmu = rng.standard_normal((Na, S))
    fs2 = np.exp(rng.standard_normal((Na, S)))
    vp = R.make_vp(rng.standard_normal((D, K)), np.exp(0.3 * rng.standard_normal(K)), np.exp(0.2 * rng.standard_normal(D)))
    vp["w"] = rng.dirichlet(np.ones(K))
    X = rng.standard_normal((Na, D))
    vln = _vbmc_lnpdf(vp, X)
    assert np.allclose(vln, np.log(R.vbmc_pdf_transformed(vp, X)), rtol=1e-12, atol=1e-12)
    for name, oname in (("acqviqr_vbmc", "acqviqr"), ("acqimiqr_vbmc", "acqimiqr")):
        for which in ("islogf1", "islogf2", "islogf"):
            a = np.broadcast_to(_islogf(name, which, vln, fmu, fs2), (Na, S))
            b = R.acq_islogf(oname, which, vln, fmu, fs2)
            assert np.allclose(a, b, rtol=1e-13, atol=1e-13), (name, which)
