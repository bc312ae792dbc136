"""CPU: why the path computes in fp64 (BASELINE configs[4]: "fp32 vs fp64 tolerance stress").

Measured at the noisy-path shape (D = 20, N = 800, K = 100, user-supplied s2): in float32 the GP weights
alpha = (K + diag(sn2))^-1 (y - m) are off by ~1.5e-5 relative -- already past north_star's 1e-6 -- although the kernel
matrix of this synthetic problem is perfectly conditioned (cond = 1.00002: in 20 dimensions the training points are
many length scales apart); with VBMC's default minimum noise (sn2 = 1e-5, cond ~ 1e5-1e7) float32 has no digits left.
The expected log joint VALUE, dominated by the mean-function terms, happens to survive at ~1e-7.  float64 agrees with
the oracle to 1e-12 on the same inputs.  Hence dtype "f64" everywhere on the path."""
import numpy as np

from oracle import vbmc_ref as R
from tests._cases import synth_problem


def logjoint_value(p, gp, dtype):
    """F = mean_s sum_k w_k (z_k . alpha + m0 + nu_k)  (gplogjoint.m:162-174,203), vectorised in `dtype`."""
    X = p["X"].astype(dtype)
    D = X.shape[1]
    mu, sig, lam = p["mu"].astype(dtype), p["sigma"].astype(dtype), p["lam"].astype(dtype)
    w = (np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))).astype(dtype)
    F = []
    for post in gp["post"]:
        hyp = post["hyp"]
        ell = np.exp(hyp[:D]).astype(dtype)
        ln_sf2 = dtype(2 * hyp[D])
        m0 = dtype(hyp[D + 2])
        xm = hyp[D + 3 : 2 * D + 3].astype(dtype)
        om = np.exp(hyp[2 * D + 3 : 3 * D + 3]).astype(dtype)
        alpha = post["alpha"].astype(dtype)
        tau = np.sqrt(sig[None, :] ** 2 * lam[:, None] ** 2 + ell[:, None] ** 2)                      # D x K
        lnnf = ln_sf2 + np.sum(np.log(ell)) - np.sum(np.log(tau), axis=0)                            # K
        d2 = np.sum(((mu[None, :, :] - X[:, :, None]) / tau[None, :, :]) ** 2, axis=1)               # N x K
        z = np.exp(lnnf[None, :] - dtype(0.5) * d2)
        nu = -dtype(0.5) * np.sum((mu**2 + sig[None, :] ** 2 * lam[:, None] ** 2 - 2 * mu * xm[:, None] + xm[:, None] ** 2)
                                  / om[:, None] ** 2, axis=0)
        F.append(np.sum(w * (z.T @ alpha + m0 + nu)))
    return np.mean(np.array(F, dtype=dtype))


def test_fp32_misses_the_tolerance_fp64_meets_it():
    p = synth_problem(5, 20, 800, 100, 2, noisy=True)
    gp = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=4, noisefun=p["noisefun"], s2=p["s2"])
    vp = R.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    ref = float(R.gplogjoint(vp, gp, (0, 0, 0, 0), True, True, 0)["F"])
    f64 = float(logjoint_value(p, gp, np.float64))
    f32 = float(logjoint_value(p, gp, np.float32))
    assert abs(f64 - ref) <= 1e-11 * abs(ref)
    assert 1e-9 * abs(ref) < abs(f32 - ref) < 1e-5 * abs(ref), (f32, ref)     # ~1e-7: seven digits is all float32 has
    # GP weights: Cholesky solve in each precision against the oracle's alpha
    import scipy.linalg as sla

    post, D = gp["post"][0], 20
    hyp = post["hyp"]
    err = {}
    for dt in (np.float32, np.float64):
        Xs = p["X"].astype(dt) / np.exp(hyp[:D]).astype(dt)
        sq = (Xs**2).sum(1)
        Km = dt(np.exp(2 * hyp[D])) * np.exp(-np.maximum(sq[:, None] + sq[None, :] - 2 * Xs @ Xs.T, 0) / 2)
        sn2 = R.gplite_noisefun(hyp[D + 1 : D + 1 + gp["Nnoise"]], p["X"], gp["noisefun"], p["y"], p["s2"]).astype(dt)
        m = R.gplite_meanfun(hyp[D + 1 + gp["Nnoise"] :], p["X"], 4).astype(dt)
        al = sla.cho_solve(sla.cho_factor((Km + np.diag(sn2)).astype(dt)), p["y"].astype(dt) - m)
        err[dt] = np.max(np.abs(al - post["alpha"])) / np.max(np.abs(post["alpha"]))
    assert err[np.float64] < 1e-12 and err[np.float32] > 1e-6, err           # north_star's 1e-6 is out of reach in fp32
