"""The ORACLE against the reference's own known answers (test/runtest_vbmc.m:9,17-27,120-126): the same hand-over point as
tests/test_gpu_known_answers.py -- evaluations of the test density at a design, a GP with maximum-likelihood hyper-parameters
(the oracle's gplite_nlZ + gradient), then the oracle's restatement of vpsieve_vbmc / vpoptimize_vbmc / negelcbo_vbmc / gplogjoint /
entmc_vbmc on it -- must land on the reference's lnZ and posterior mean within the reference's tolerances (0.5, 0.5).  A reduced
size (D = 2: standard deviations 1 and 2) beside the reference's own test 1 (D = 6), both with K = 2: seconds of pure NumPy on one core.  This is the
one check of the oracle that is anchored on numbers the REFERENCE holds rather than on vectors computed for this repository."""
import numpy as np
from scipy.optimize import minimize

from oracle import vbmc_ref as R

TOLERR = (0.5, 0.5)              # test/runtest_vbmc.m:9


def target(x):                   # test/runtest_vbmc.m:26
    i = np.arange(1, x.shape[1] + 1)
    return np.sum(-0.5 * (x / i) ** 2, axis=1) - np.sum(np.log(i)) - 0.5 * x.shape[1] * np.log(2 * np.pi)


import pytest


@pytest.mark.parametrize("D,npost,nbox", [(2, 36, 12), (6, 100, 50)])       # D = 6 is test 1 of runtest_vbmc.m itself
def test_oracle_pipeline_returns_the_reference_known_answers(D, npost, nbox):
    rng = np.random.default_rng(3)
    K = 2
    sd = np.arange(1, D + 1, dtype=np.float64)
    X = np.concatenate([rng.standard_normal((npost, D)) * 1.2 * sd, rng.uniform(-2 * D, 2 * D, size=(nbox, D)), -np.ones((1, D))], axis=0)
    y = target(X)
    N = X.shape[0]
    gp0 = {"X": X, "y": y, "s2": None, "covfun": 1, "Ncov": D + 1, "noisefun": (1, 0, 0), "Nnoise": 1, "meanfun": 4, "Nmean": 2 * D + 1,
           "meanfun_extras": None, "intmeanfun": 0}
    h0 = np.concatenate([np.log(np.std(X, axis=0)), [np.log(np.std(y))], [np.log(1e-2)], [np.max(y)], np.mean(X, axis=0), np.log(np.std(X, axis=0))])
    lb = np.concatenate([h0[:D] - 4, [h0[D] - 6], [np.log(1e-4)], [np.max(y) - 10 * np.ptp(y)], np.min(X, axis=0), h0[D + 3 + D:] - 3])
    ub = np.concatenate([h0[:D] + 4, [h0[D] + 6], [np.log(2.0)], [np.max(y) + 10 * np.ptp(y)], np.max(X, axis=0), h0[D + 3 + D:] + 3])

    def f(h):
        nlz, g = R.gplite_nlZ(h, gp0, None, True)
        return float(nlz), np.asarray(g, dtype=np.float64).reshape(-1)

    r = minimize(f, h0, jac=True, method="L-BFGS-B", bounds=list(zip(lb, ub)), options={"maxiter": 200})
    gp = R.gplite_post(r.x.reshape(-1, 1), X, y, meanfun=4)
    order = np.argsort(-y)
    lam = np.std(X[order[:max(20, npost // 2)]], axis=0)
    vp = R.make_vp(X[order[:K]].T.copy(), np.full(K, 0.6), lam * np.sqrt(D / np.sum(lam ** 2)))
    vp["w"] = np.full(K, 1.0 / K)
    draws = np.random.default_rng(7)

    opts = {"MaxIterStochastic": 250, "NSentFine": lambda k: 256 * k, "NSent": lambda k: 40 * k}

    def eps_for(kind, slot, it, K_, Ns):
        if Ns is None:                                       # the full-ELCBO evaluations: NSentFine / K per component (eval_fullelcbo)
            Ns = int(np.ceil(opts["NSentFine"](K_) / K_))
        return draws.standard_normal((K_, (Ns + 1) // 2, D))    # MATLAB's randn stream of the reference, here NumPy's

    for it in range(2):
        vp, _, _ = R.vpoptimize_vbmc(8 if it == 0 else 4, 1, vp, gp, options=opts, rng=np.random.default_rng(it), eps_for=eps_for)
    st = vp["stats"]
    vmu = np.asarray(vp["mu"]).reshape(D, -1) @ np.asarray(vp["w"]).reshape(-1)
    err = (abs(st["elbo"] - 0.0), float(np.sqrt(np.mean(vmu ** 2))))                     # test/runtest_vbmc.m:120-126
    assert err[0] < TOLERR[0] and err[1] < TOLERR[1], (err, st["elbo"], vmu)
