"""CPU oracle (NumPy) for the VBMC ELBO inner loop -- TEST INFRASTRUCTURE ONLY.

This module is a line-by-line NumPy restatement of the MATLAB reference
(acerbilab/vbmc v1.0.12) for the hot path named in BASELINE.json and its SURVEY 8(f) neighbours (gplite_nlZ, the
acquisition functions).  It is the
*checker*: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it.  The product path (``vbmc_amd``) never does.

PARITY UNPINNED (by the reference): the reference is pure MATLAB, it holds no
function-level golden vectors for this path and neither
MATLAB nor Octave exists in the build container, so the reference cannot be run
to generate vectors.  What pins this restatement instead is committed under
``tests/golden/`` and exercised by ``tests/test_oracle_*.py``:
  * the reference's only known answers -- test/runtest_vbmc.m: |ELBO - lnZ| < 0.5 and RMSE of the posterior mean < 0.5 on its
    test density 1 -- required of this module's whole pipeline from the GP fit on (``tests/test_oracle_known_answers.py``),
  * 50-digit mpmath re-evaluation of the same formulas (``oracle/mp_golden.py``),
  * closed forms (K=1 entropy, alpha=0 log-joint, identical-component bounds),
  * finite differences where the reference's gradient is an exact derivative
    (all of gplogjoint, entlb; the eta block of entmc),
  * linear-algebra identities for gplite_post / gplite_pred, rank-1 == full,
  * for the rows added after the ELBO path: gplite_nlZ against the 50-digit definition and 50-digit
    central differences (``tests/golden/mp_nlz_case*.json``); the IQR acquisition's look-ahead variance against an
    actual rank-one update; vbmc_pdf and gplite_hypprior against SciPy.
``tools/dump_golden.m`` + ``tools/compare_matlab_golden.py`` let anyone with MATLAB pin the vectors to the reference.

Every function cites the reference file:line it follows (paths relative to the
reference checkout).  MATLAB shapes are kept: ``mu`` is D x K, ``sigma`` (K,),
``lambda_`` (D,), ``w`` (K,), ``eta`` (K,); ``gp['X']`` is N x D.  All fp64.

The implicit ``randn`` stream of ``entmc_vbmc`` is an explicit input here:
``eps`` has shape (K, M/2, D) in C order, which is byte-identical to K
consecutive MATLAB ``randn(D,1,M/2)`` blocks (d fastest, then sample, then j).
"""
from __future__ import annotations

import copy
import math

import numpy as np

EPS = np.finfo(np.float64).eps  # MATLAB `eps`
REALMIN = np.finfo(np.float64).tiny

# --------------------------------------------------------------------------
# small containers
# --------------------------------------------------------------------------


def make_vp(mu, sigma, lambda_, w=None, eta=None, optimize=(True, True, True, True), delta=None):
    """Variational posterior struct (misc/setupvars_vbmc.m:78-99)."""
    mu = np.array(mu, dtype=np.float64)
    D, K = mu.shape
    vp = {
        "D": D,
        "K": K,
        "mu": mu,
        "sigma": np.array(sigma, dtype=np.float64).reshape(K),
        "lambda": np.array(lambda_, dtype=np.float64).reshape(D),
        "w": (np.full(K, 1.0 / K) if w is None else np.array(w, dtype=np.float64).reshape(K)),
        "optimize_mu": bool(optimize[0]),
        "optimize_sigma": bool(optimize[1]),
        "optimize_lambda": bool(optimize[2]),
        "optimize_weights": bool(optimize[3]),
        "delta": delta,
        "bounds": None,
        "stats": None,
    }
    if eta is not None:
        vp["eta"] = np.array(eta, dtype=np.float64).reshape(K)
    return vp


def copy_vp(vp):
    return copy.deepcopy(vp)


# --------------------------------------------------------------------------
# sq_dist  (utils/sq_dist.m:14-50, identical copy in gplite/private/sq_dist.m)
# --------------------------------------------------------------------------


def sq_dist(a, b=None):
    """Pairwise squared distances between columns of a (D x n) and b (D x m)."""
    a = np.asarray(a, dtype=np.float64)
    D, n = a.shape
    if b is None:
        mu = a.mean(axis=1, keepdims=True)  # sq_dist.m:26
        a = a - mu
        b = a
        m = n
    else:
        b = np.asarray(b, dtype=np.float64)
        d, m = b.shape
        if d != D:
            raise ValueError("Error: column lengths must agree.")
        mu = (m / (n + m)) * b.mean(axis=1, keepdims=True) + (n / (n + m)) * a.mean(axis=1, keepdims=True)  # :36
        a = a - mu
        b = b - mu
    C = np.sum(a * a, axis=0)[:, None] + (np.sum(b * b, axis=0)[None, :] - 2.0 * (a.T @ b))  # :45
    return np.maximum(C, 0.0)  # :49


# --------------------------------------------------------------------------
# GP mean / noise function tables (only what VBMC's defaults reach)
# --------------------------------------------------------------------------


def gplite_meanfun(hyp_mean, X, meanfun):
    """gplite/gplite_meanfun.m:398-436, ids 0 (zero), 1 (const), 4 (negquad)."""
    X = np.asarray(X, dtype=np.float64)
    N, D = X.shape
    if meanfun == 0:
        return np.zeros(N)
    if meanfun == 1:
        return hyp_mean[0] * np.ones(N)
    if meanfun == 4:
        m0 = hyp_mean[0]
        xm = hyp_mean[1 : D + 1]
        omega = np.exp(hyp_mean[D + 1 : 2 * D + 1])
        z2 = ((X - xm[None, :]) / omega[None, :]) ** 2  # :430
        return m0 - 0.5 * np.sum(z2, axis=1)  # :431
    raise NotImplementedError("meanfun %r outside the hot path" % (meanfun,))


def meanfun_nhyp(meanfun, D):
    return {0: 0, 1: 1, 4: 2 * D + 1}[meanfun]


def gplite_noisefun(hyp_noise, X, noisefun, y=None, s2=None):
    """gplite/gplite_noisefun.m:176-210."""
    idx = 0
    if noisefun[0] == 0:
        sn2 = EPS
    else:
        sn2 = math.exp(2.0 * hyp_noise[idx])  # :181
        idx += 1
    if s2 is None or np.size(s2) == 0:
        s2 = 0.0  # :51  (empty s2 counts as zero, e.g. gplite_pred without s2star)
    if noisefun[1] == 1:
        sn2 = sn2 + np.asarray(s2, dtype=np.float64)  # :188
    elif noisefun[1] == 2:
        sn2 = sn2 + math.exp(hyp_noise[idx]) * np.asarray(s2, dtype=np.float64)  # :190
        idx += 1
    if len(noisefun) > 2 and noisefun[2] == 1:
        if y is not None and np.size(y) > 0:
            ythresh = hyp_noise[idx]
            w2 = math.exp(2.0 * hyp_noise[idx + 1])
            zz = np.maximum(0.0, ythresh - np.asarray(y, dtype=np.float64))
            sn2 = sn2 + w2 * zz**2  # :202
        idx += 2
    return sn2


def noisefun_nhyp(noisefun):
    n = 0
    if noisefun[0] == 1:
        n += 1
    if noisefun[1] == 2:
        n += 1
    if len(noisefun) > 2 and noisefun[2] == 1:
        n += 2
    return n


# --------------------------------------------------------------------------
# dense helpers standing in for MATLAB built-ins (chol, \ on triangular)
# --------------------------------------------------------------------------


def chol_upper(A):
    """MATLAB ``[R,p] = chol(A)``: upper R with R'R = A; p>0 on failure."""
    A = np.array(A, dtype=np.float64)
    n = A.shape[0]
    R = np.zeros_like(A)
    for j in range(n):
        s = A[j, j] - np.dot(R[:j, j], R[:j, j])
        if not (s > 0.0) or not np.isfinite(s):
            return R, j + 1
        R[j, j] = math.sqrt(s)
        if j + 1 < n:
            R[j, j + 1 :] = (A[j, j + 1 :] - R[:j, j] @ R[:j, j + 1 :]) / R[j, j]
    return R, 0


def solve_upper(R, B):
    """R \\ B with R upper triangular (back substitution)."""
    from scipy.linalg import solve_triangular

    return solve_triangular(R, B, lower=False, trans="N", check_finite=False)


def solve_upper_t(R, B):
    """R' \\ B with R upper triangular (forward substitution on R')."""
    from scipy.linalg import solve_triangular

    return solve_triangular(R, B, lower=False, trans="T", check_finite=False)


# --------------------------------------------------------------------------
# gplite_core (no-nlZ branch) and gplite_post
# --------------------------------------------------------------------------


def gplite_core_post(hyp, gp):
    """gplite/private/gplite_core.m:1-102,278-291 with compute_nlZ = 0.

    Returns the ``post`` struct {hyp, alpha, sW, L, sn2_mult, Lchol}.
    Only covfun 1 (SE-ARD), no output warping, no integrated mean.
    """
    X = gp["X"]
    y = gp["y"]
    N, D = X.shape
    Ncov, Nnoise, Nmean = gp["Ncov"], gp["Nnoise"], gp["Nmean"]
    hyp = np.asarray(hyp, dtype=np.float64)
    hyp_noise = hyp[Ncov : Ncov + Nnoise]
    sn2 = gplite_noisefun(hyp_noise, X, gp["noisefun"], y, gp.get("s2"))  # :38
    sn2_mult = 1.0  # :40
    hyp_mean = hyp[Ncov + Nnoise : Ncov + Nnoise + Nmean]
    m = gplite_meanfun(hyp_mean, X, gp["meanfun"])  # :46

    ell = np.exp(hyp[0:D])  # :53
    sf2 = math.exp(2.0 * hyp[D])  # :54
    K_mat = sq_dist(X.T / ell[:, None])  # :55
    K_mat = sf2 * np.exp(-K_mat / 2.0)  # :56

    Lchol = bool(np.min(sn2) >= 1e-6)  # :67
    if Lchol:
        if np.isscalar(sn2) or np.ndim(sn2) == 0:
            sn2div = float(sn2)
            sn2_mat = np.eye(N)
        else:
            sn2div = float(np.min(sn2))
            sn2_mat = np.diag(sn2 / sn2div)
        for _ in range(10):  # :77-80
            L, p = chol_upper(K_mat / (sn2div * sn2_mult) + sn2_mat)
            if p > 0:
                sn2_mult *= 10.0
            else:
                break
        sl = sn2div * sn2_mult
        pL = L
    else:
        if np.isscalar(sn2) or np.ndim(sn2) == 0:
            sn2_mat = float(sn2) * np.eye(N)
        else:
            sn2_mat = np.diag(sn2)
        for _ in range(10):  # :91-94
            L, p = chol_upper(K_mat + sn2_mult * sn2_mat)
            if p > 0:
                sn2_mult *= 10.0
            else:
                break
        sl = 1.0
        pL = -solve_upper(L, solve_upper_t(L, np.eye(N)))  # :98
    alpha = solve_upper(L, solve_upper_t(L, y - m)) / sl  # :102
    post = {
        "hyp": hyp.copy(),
        "alpha": alpha,
        "sW": np.ones(N) / math.sqrt(float(np.min(sn2)) * sn2_mult),  # :281
        "L": pL,
        "sn2_mult": sn2_mult,
        "Lchol": Lchol,
    }
    return post


def gplite_meanfun_grad(hyp_mean, X, meanfun):
    """gplite/gplite_meanfun.m:395-436 with compute_grad: (m, dm) with dm N x Nmean (None for id 0)."""
    X = np.asarray(X, dtype=np.float64)
    N, D = X.shape
    m = gplite_meanfun(hyp_mean, X, meanfun)
    if meanfun == 0:
        return m, None  # :402
    if meanfun == 1:
        return m, np.ones((N, 1))  # :406
    xm = hyp_mean[1 : D + 1]
    omega = np.exp(hyp_mean[D + 1 : 2 * D + 1])
    z2 = ((X - xm[None, :]) / omega[None, :]) ** 2
    dm = np.zeros((N, 2 * D + 1))
    dm[:, 0] = 1.0  # :433
    dm[:, 1 : D + 1] = (X - xm[None, :]) / omega[None, :] ** 2  # :434 with sgn = -1
    dm[:, D + 1 :] = z2  # :435
    return m, dm


def gplite_noisefun_grad(hyp_noise, X, noisefun, y=None, s2=None):
    """gplite/gplite_noisefun.m:164-210 with compute_grad: (sn2, dsn2); dsn2 is 1 x Nnoise for a
    constant noise model and N x Nnoise otherwise (:166-171)."""
    N = np.asarray(X).shape[0]
    Nnoise = noisefun_nhyp(noisefun)
    vector = any(v > 0 for v in noisefun[1:])
    dsn2 = np.zeros((N if vector else 1, Nnoise))
    idx = 0
    if noisefun[0] == 0:
        sn2 = EPS
    else:
        sn2 = math.exp(2.0 * hyp_noise[idx])
        dsn2[:, idx] = 2.0 * sn2  # :182
        idx += 1
    if s2 is None or np.size(s2) == 0:
        s2 = 0.0  # :51
    if noisefun[1] == 1:
        sn2 = sn2 + np.asarray(s2, dtype=np.float64)
    elif noisefun[1] == 2:
        sn2 = sn2 + math.exp(hyp_noise[idx]) * np.asarray(s2, dtype=np.float64)
        dsn2[:, idx] = math.exp(hyp_noise[idx]) * np.asarray(s2, dtype=np.float64)  # :192
        idx += 1
    if len(noisefun) > 2 and noisefun[2] == 1:
        if y is not None and np.size(y) > 0:
            yv = np.asarray(y, dtype=np.float64)
            ythresh = hyp_noise[idx]
            w2 = math.exp(2.0 * hyp_noise[idx + 1])
            zz = np.maximum(0.0, ythresh - yv)
            sn2 = sn2 + w2 * zz**2
            dsn2[:, idx] = 2.0 * w2 * (ythresh - yv) * (zz > 0)  # :204
            dsn2[:, idx + 1] = 2.0 * w2 * zz**2  # :205
        idx += 2
    return sn2, dsn2


def gplite_core_nlZ(hyp, gp, compute_grad=True):
    """gplite/private/gplite_core.m:1-102 + the compute_nlZ branch :128-275 (covfun 1, no output
    warping, no integrated mean): (nlZ, dnlZ).  A Cholesky that still fails after the 10
    noise-inflation retries makes MATLAB error downstream; the caller (gplite_train.m:542-546) maps
    that to NaN, and so does this function."""
    X = gp["X"]
    y = gp["y"]
    N, D = X.shape
    Ncov, Nnoise, Nmean = gp["Ncov"], gp["Nnoise"], gp["Nmean"]
    hyp = np.asarray(hyp, dtype=np.float64).reshape(-1)
    Nhyp = hyp.size
    sn2, dsn2 = gplite_noisefun_grad(hyp[Ncov : Ncov + Nnoise], X, gp["noisefun"], y, gp.get("s2"))  # :36
    sn2_mult = 1.0
    m, dm = gplite_meanfun_grad(hyp[Ncov + Nnoise : Ncov + Nnoise + Nmean], X, gp["meanfun"])  # :44
    ell = np.exp(hyp[0:D])
    sf2 = math.exp(2.0 * hyp[D])
    K_mat = sf2 * np.exp(-sq_dist(X.T / ell[:, None]) / 2.0)  # :55-56
    scalar = np.ndim(sn2) == 0
    Lchol = bool(np.min(sn2) >= 1e-6)
    p = 1
    if Lchol:
        sn2div = float(sn2) if scalar else float(np.min(sn2))
        sn2_mat = np.eye(N) if scalar else np.diag(sn2 / sn2div)
        for _ in range(10):
            L, p = chol_upper(K_mat / (sn2div * sn2_mult) + sn2_mat)
            if p > 0:
                sn2_mult *= 10.0
            else:
                break
        sl = sn2div * sn2_mult
    else:
        sn2_mat = float(sn2) * np.eye(N) if scalar else np.diag(sn2)
        for _ in range(10):
            L, p = chol_upper(K_mat + sn2_mult * sn2_mat)
            if p > 0:
                sn2_mult *= 10.0
            else:
                break
        sl = 1.0
    if p > 0:
        return float("nan"), np.full(Nhyp, np.nan)
    alpha = solve_upper(L, solve_upper_t(L, y - m)) / sl  # :102
    nlZ = float((y - m) @ alpha / 2.0 + np.sum(np.log(np.diag(L))) + N * math.log(2 * math.pi * sl) / 2.0)  # :205
    if not compute_grad:
        return nlZ, None
    dnlZ = np.zeros(Nhyp)
    Q = solve_upper(L, solve_upper_t(L, np.eye(N))) / sl - np.outer(alpha, alpha)  # :240
    for i in range(D):  # :244-247
        K_temp = K_mat * sq_dist(X[:, i][None, :] / ell[i])
        dnlZ[i] = np.sum(Q * K_temp) / 2.0
    dnlZ[D] = np.sum(Q * (2.0 * K_mat)) / 2.0  # :248
    if scalar:  # :257-259
        trQ = np.trace(Q)
        for i in range(Nnoise):
            dnlZ[Ncov + i] = 0.5 * sn2_mult * dsn2[0, i] * trQ
    else:  # :261-262
        dgQ = np.diag(Q)
        for i in range(Nnoise):
            dnlZ[Ncov + i] = 0.5 * sn2_mult * np.sum(dsn2[:, i] * dgQ)
    if Nmean > 0:
        dnlZ[Ncov + Nnoise : Ncov + Nnoise + Nmean] = -dm.T @ alpha  # :274
    return nlZ, dnlZ


def gplite_hypprior(hyp, hprior, compute_grad=True):
    """gplite/gplite_hypprior.m:17-65: (lp, dlp) of independent uniform / Gaussian / Student-t priors."""
    from math import lgamma

    hyp = np.asarray(hyp, dtype=np.float64).reshape(-1)
    Nhyp = hyp.size
    mu = np.asarray(hprior["mu"], dtype=np.float64).reshape(-1)
    sigma = np.abs(np.asarray(hprior["sigma"], dtype=np.float64).reshape(-1))
    df = hprior.get("df")
    df = 7.0 * np.ones(Nhyp) if df is None or np.size(df) == 0 else np.asarray(df, dtype=np.float64).reshape(-1)  # :32-36
    uidx = ~np.isfinite(mu) | ~np.isfinite(sigma)
    gidx = ~uidx & ((df == 0) | ~np.isfinite(df)) & np.isfinite(sigma)
    tidx = ~uidx & (df > 0) & np.isfinite(df)
    z2 = np.zeros(Nhyp)
    gt = gidx | tidx
    z2[gt] = ((hyp[gt] - mu[gt]) / sigma[gt]) ** 2
    lp = 0.0
    dlp = np.zeros(Nhyp)
    if np.any(gidx):  # :47-52
        lp -= 0.5 * np.sum(np.log(2 * math.pi * sigma[gidx] ** 2) + z2[gidx])
        dlp[gidx] = -(hyp[gidx] - mu[gidx]) / sigma[gidx] ** 2
    if np.any(tidx):  # :55-62
        d = df[tidx]
        lg = np.array([lgamma(0.5 * (v + 1)) - lgamma(0.5 * v) for v in d])
        lp += np.sum(lg - 0.5 * np.log(math.pi * d) - np.log(sigma[tidx]) - 0.5 * (d + 1) * np.log1p(z2[tidx] / d))
        dlp[tidx] = -(d + 1) / d / (1 + z2[tidx] / d) * (hyp[tidx] - mu[tidx]) / sigma[tidx] ** 2
    return (float(lp), dlp) if compute_grad else (float(lp), None)


def gplite_nlZ(hyp, gp, hprior=None, compute_grad=True):
    """gplite/gplite_nlZ.m:1-72: negative log marginal likelihood (minus the log hyper-prior) and gradient."""
    hyp = np.asarray(hyp, dtype=np.float64)
    if hyp.ndim == 2 and hyp.shape[1] > 1 and compute_grad:
        raise ValueError("gplite_nlZ:NoSampling")  # :41-44
    hyp = hyp.reshape(-1)
    if hyp.size != gp["Ncov"] + gp["Nnoise"] + gp["Nmean"]:
        raise ValueError("gplite_nlZ:dimmismatch")  # :38-40
    nlZ, dnlZ = gplite_core_nlZ(hyp, gp, compute_grad)
    if hprior is not None:  # :58-68
        P, dP = gplite_hypprior(hyp, hprior, compute_grad)
        nlZ = nlZ - P
        if compute_grad:
            dnlZ = dnlZ - dP
    return nlZ, dnlZ


def gplite_post(hyp, X, y, meanfun=4, noisefun=(1, 0, 0), s2=None):
    """gplite/gplite_post.m:94-172 (fresh GP struct + full posterior per sample).

    ``hyp`` is Nhyp x S (columns are hyper-parameter samples).
    """
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    N, D = X.shape
    hyp = np.asarray(hyp, dtype=np.float64)
    if hyp.ndim == 1:
        hyp = hyp[:, None]
    gp = {
        "X": X,
        "y": y,
        "s2": None if s2 is None else np.asarray(s2, dtype=np.float64).reshape(-1),
        "covfun": 1,
        "Ncov": D + 1,
        "noisefun": tuple(noisefun),
        "Nnoise": noisefun_nhyp(noisefun),
        "meanfun": meanfun,
        "Nmean": meanfun_nhyp(meanfun, D),
        "meanfun_extras": None,
        "intmeanfun": 0,
    }
    if hyp.shape[0] != gp["Ncov"] + gp["Nnoise"] + gp["Nmean"]:
        raise ValueError("gplite_post:dimmismatch")  # :152-155
    gp["post"] = [gplite_core_post(hyp[:, s], gp) for s in range(hyp.shape[1])]  # :167-172
    return gp


def gplite_post_rank1(gp, xstar, ystar):
    """gplite/gplite_post.m:173-251: rank-1 append of one training point."""
    gp = copy.deepcopy(gp)
    xstar = np.asarray(xstar, dtype=np.float64).reshape(1, -1)
    ystar = float(ystar)
    if gp.get("s2") is not None:
        raise NotImplementedError("rank-1 with s2 falls back to full update (:76-79)")
    N, D = gp["X"].shape
    Ncov, Nnoise = gp["Ncov"], gp["Nnoise"]
    mstar, vstar, _, _ = gplite_pred(gp, xstar, np.array([ystar]), None, ssflag=True)  # :189
    for s, post in enumerate(gp["post"]):
        hyp = post["hyp"]
        hyp_noise = hyp[Ncov : Ncov + Nnoise]
        sn2 = gplite_noisefun(hyp_noise, xstar, gp["noisefun"], np.array([ystar]), None)
        sn2 = float(np.min(sn2)) if np.ndim(sn2) else float(sn2)
        sn2_eff = sn2 * post["sn2_mult"]  # :207
        ell = np.exp(hyp[0:D])
        sf2 = math.exp(2.0 * hyp[D])
        Kss = sf2  # :213
        Ks_mat = sq_dist(gp["X"].T / ell[:, None], xstar.T / ell[:, None])  # :214
        Ks_mat = sf2 * np.exp(-Ks_mat / 2.0)
        L = post["L"]
        if post["Lchol"]:
            alpha_update = solve_upper(L, solve_upper_t(L, Ks_mat)) / sn2_eff  # :227
            new_col = solve_upper_t(L, Ks_mat) / sn2_eff  # :228-229
            newL = np.zeros((N + 1, N + 1))
            newL[:N, :N] = L
            newL[:N, N:] = new_col
            newL[N, N] = math.sqrt(1.0 + Kss / sn2_eff - float((new_col.T @ new_col)[0, 0]))  # :232
        else:
            alpha_update = -L @ Ks_mat  # :234
            v = -alpha_update / vstar[0, s]
            newL = np.zeros((N + 1, N + 1))
            newL[:N, :N] = L + v @ alpha_update.T
            newL[:N, N:] = -v
            newL[N:, :N] = -v.T
            newL[N, N] = -1.0 / vstar[0, s]  # :236
        post["L"] = newL
        post["sW"] = np.concatenate([post["sW"], [1.0 / math.sqrt(sn2_eff)]])  # :239
        upd = np.concatenate([alpha_update[:, 0], [-1.0]])
        post["alpha"] = np.concatenate([post["alpha"], [0.0]]) + (mstar[0, s] - ystar) / vstar[0, s] * upd  # :242-244
    gp["X"] = np.vstack([gp["X"], xstar])
    gp["y"] = np.concatenate([gp["y"], [ystar]])
    return gp


# --------------------------------------------------------------------------
# gplite_pred
# --------------------------------------------------------------------------


def gplite_pred(gp, Xstar, ystar=None, s2star=None, ssflag=False, nargout=4):
    """gplite/gplite_pred.m:1-165 -> (ymu, ys2, fmu, fs2); Nstar x S if ssflag
    else Nstar x 1 vectors (returned as (Nstar,S) / (Nstar,) arrays)."""
    X = gp["X"]
    N, D = X.shape
    S = len(gp["post"])
    Xstar = np.asarray(Xstar, dtype=np.float64)
    Nstar = Xstar.shape[0]
    Ncov, Nnoise, Nmean = gp["Ncov"], gp["Nnoise"], gp["Nmean"]
    fmu = np.zeros((Nstar, S))
    ymu = np.zeros((Nstar, S))
    fs2 = np.zeros((Nstar, S))
    ys2 = np.zeros((Nstar, S))
    for s, post in enumerate(gp["post"]):
        hyp = post["hyp"]
        alpha, L, Lchol, sW, sn2_mult = post["alpha"], post["L"], post["Lchol"], post["sW"], post["sn2_mult"]
        hyp_noise = hyp[Ncov : Ncov + Nnoise]
        sn2_star = gplite_noisefun(hyp_noise, Xstar, gp["noisefun"], ystar, s2star)  # :63
        hyp_mean = hyp[Ncov + Nnoise : Ncov + Nnoise + Nmean]
        mstar = gplite_meanfun(hyp_mean, Xstar, gp["meanfun"])  # :67
        ell = np.exp(hyp[0:D])
        sf2 = math.exp(2.0 * hyp[D])
        Ks_mat = sq_dist(X.T / ell[:, None], Xstar.T / ell[:, None])  # :73
        Ks_mat = sf2 * np.exp(-Ks_mat / 2.0)
        kss = sf2 * np.ones(Nstar)  # :75
        fmu[:, s] = mstar + Ks_mat.T @ alpha  # :83
        ymu[:, s] = fmu[:, s]
        if Lchol:
            V = solve_upper_t(L, sW[:, None] * Ks_mat)  # :99
            fs2[:, s] = kss - np.sum(V * V, axis=0)  # :100
        else:
            LKs = L @ Ks_mat
            fs2[:, s] = kss + np.sum(Ks_mat * LKs, axis=0)  # :103
        fs2[:, s] = np.maximum(fs2[:, s], 0.0)  # :120
        ys2[:, s] = fs2[:, s] + sn2_star * sn2_mult  # :121
    lp = None
    if ystar is not None and np.size(ystar) > 0 and nargout > 4:  # :124-127 (per hyper-sample, never averaged)
        ys_ = np.asarray(ystar, dtype=np.float64).reshape(-1, 1)
        lp = -0.5 * (ys_ - ymu) ** 2 / ys2 - 0.5 * np.log(2 * math.pi * ys2)
    if nargout > 4:
        avg = gplite_pred(gp, Xstar, ystar, s2star, ssflag)
        return tuple(avg) + (lp,)
    if S > 1 and not ssflag:  # :154-165
        fbar = np.sum(fmu, axis=1) / S
        ybar = np.sum(ymu, axis=1) / S
        vf = np.sum((fmu - fbar[:, None]) ** 2, axis=1) / (S - 1)
        fs2 = np.sum(fs2, axis=1) / S + vf
        vy = np.sum((ymu - ybar[:, None]) ** 2, axis=1) / (S - 1)
        ys2 = np.sum(ys2, axis=1) / S + vy
        return ybar, ys2, fbar, fs2
    if not ssflag:
        return ymu[:, 0], ys2[:, 0], fmu[:, 0], fs2[:, 0]
    return ymu, ys2, fmu, fs2


# --------------------------------------------------------------------------
# softmax Jacobian used by gplogjoint / entmc / entlb / negelcbo
# --------------------------------------------------------------------------


REALMIN = 2.2250738585072014e-308
REALMAX = 1.7976931348623157e308


def vbmc_pdf_transformed(vp, X):
    """vbmc_pdf(vp,X,0) (vbmc_pdf.m:38-71): mixture density in the transformed space, Gaussian components."""
    X = np.asarray(X, dtype=np.float64)
    N, D = X.shape
    lam = np.asarray(vp["lambda"], dtype=np.float64).reshape(-1)
    nf = 1.0 / (2 * math.pi) ** (D / 2.0) / np.prod(lam)  # :58
    y = np.zeros(N)
    for k in range(vp["K"]):  # :60-63
        d2 = np.sum(((X - vp["mu"][:, k][None, :]) / (vp["sigma"][k] * lam[None, :])) ** 2, axis=1)
        y = y + nf * vp["w"][k] / vp["sigma"][k] ** D * np.exp(-0.5 * d2)
    return y


def _acq_sqdist_rows(a, b):
    """Local sq_dist of acq/acqfsn2_vbmc.m:24-32 (points in ROWS, joint-mean centring)."""
    n, m = a.shape[0], b.shape[0]
    mu = (m / (n + m)) * np.mean(b, axis=0) + (n / (n + m)) * np.mean(a, axis=0)
    a = a - mu
    b = b - mu
    C = np.sum(a * a, axis=1)[:, None] + (np.sum(b * b, axis=1)[None, :] - 2 * a @ b.T)
    return np.maximum(C, 0.0)


def acq_function(name, Xs, vp, gp, optimState, fmu, fs2, fbar, vtot):
    """acq/acqf_vbmc.m:6-10, acq/acqflog_vbmc.m:14-18, acq/acqus_vbmc.m:6-9, acq/acqfsn2_vbmc.m:6-17."""
    p = np.maximum(vbmc_pdf_transformed(vp, Xs), REALMIN)
    z = optimState.get("ymax", 0.0)
    if name == "acqf":
        return -vtot * np.exp(fbar - z) * p
    if name == "acqflog":
        return -(np.log(vtot) + fbar - z + np.log(p))
    if name == "acqus":
        return -vtot * p**2
    if name == "acqfsn2":
        pos = np.argmin(_acq_sqdist_rows(Xs / optimState["gplengthscale"][None, :], gp["X_rescaled"]), axis=1)
        sn2 = np.asarray(gp["sn2new"], dtype=np.float64)[pos]
        return -vtot * (1 - sn2 / (vtot + sn2)) * np.exp(fbar - z) * p
    raise NotImplementedError(name)


def acq_is_precompute(gp, Xa):
    r"""private/activeimportancesampling_vbmc.m:248-276 (Step 3): cross-kernel of the importance points with the
    training inputs and Ctmp = (L\(L'\Kax'))/sn2_eff (Lchol) or L*Kax' per hyper-sample.  Xa: Na x D, or
    Na x D x S for per-hyper-sample inputs.  Returns (Kax_mat Na x N x S, Ctmp_mat N x Na x S)."""
    X = gp["X"]
    N, D = X.shape
    S = len(gp["post"])
    Xa = np.asarray(Xa, dtype=np.float64)
    Na = Xa.shape[0]
    Kax = np.zeros((Na, N, S))
    Ctmp = np.zeros((N, Na, S))
    for s, post in enumerate(gp["post"]):
        xa = Xa if Xa.ndim == 2 else Xa[:, :, s]
        hyp = post["hyp"]
        ell = np.exp(hyp[0:D])
        sf2 = math.exp(2.0 * hyp[D])
        Kax[:, :, s] = sf2 * np.exp(-_acq_sqdist_rows(xa / ell[None, :], X / ell[None, :]) / 2.0)  # :264-265
        L = post["L"]
        if post["Lchol"]:
            sn2_eff = 1.0 / post["sW"][0] ** 2
            Ctmp[:, :, s] = solve_upper(L, solve_upper_t(L, Kax[:, :, s].T)) / sn2_eff  # :271
        else:
            Ctmp[:, :, s] = L @ Kax[:, :, s].T  # :273
    return Kax, Ctmp


ACQ_U = 0.6745  # norminv(0.75), acqviqr_vbmc.m:4 / acqimiqr_vbmc.m:4


def acq_islogf(name, which, vlnpdf, fmu, fs2):
    """acqfun('islogf1' | 'islogf2' | 'islogf', vlnpdf, [], [], fmu, fs2): acq/acqviqr_vbmc.m:13-30 (name 'acqviqr') and
    acq/acqimiqr_vbmc.m:12-27 ('acqimiqr').  fmu, fs2: Na x S."""
    fs2 = np.asarray(fs2, dtype=np.float64)
    out = np.zeros_like(fs2)
    for i in range(fs2.shape[0]):
        for s in range(fs2.shape[1]):
            fs = math.sqrt(fs2[i, s])
            added = ACQ_U * fs + math.log1p(-math.exp(-2 * ACQ_U * fs))
            if name == "acqviqr":
                fixed = 0.0 if which == "islogf1" else float(np.asarray(vlnpdf).reshape(-1)[i])      # :19 zeros(size(fs2)); :29 vp + ...
            else:
                fixed = float(np.asarray(fmu)[i, s])                                                  # acqimiqr :16, :26
            out[i, s] = fixed if which == "islogf1" else (added if which == "islogf2" else fixed + added)
    return out


def activesample_proposalpdf(Xa, gp, vp_is, w_vp, rect_delta, name, vp, isamplevp_flag):
    """[lnw,fs2] = activesample_proposalpdf(Xa,gp,vp_is,w_vp,rect_delta,acqfun,vp,isamplevp_flag)
    (private/activeimportancesampling_vbmc.m:301-340), point by point.  Returns lnw (Na x S) and fs2 (Na x S)."""
    X = gp["X"]
    N, D = X.shape
    Xa = np.asarray(Xa, dtype=np.float64)
    Na = Xa.shape[0]
    pr = gplite_pred(gp, Xa, None, None, True)                                      # :307
    S = len(gp["post"])
    fmu = np.asarray(pr[2]).reshape(Na, S)
    fs2 = np.asarray(pr[3]).reshape(Na, S)
    templ = np.full((Na, 1 + N), -np.inf)
    if w_vp > 0:                                                                     # :313-318
        with np.errstate(divide="ignore"):
            templ[:, 0] = np.log(vbmc_pdf_transformed(vp_is, Xa)) + math.log(w_vp)
    if isamplevp_flag:                                                               # :321-326
        with np.errstate(divide="ignore"):
            vln = np.maximum(np.log(np.maximum(vbmc_pdf_transformed(vp, Xa), 0.0)), math.log(REALMIN))
        lny = acq_islogf(name, "islogf1", vln, fmu, fs2)
    else:
        lny = acq_islogf(name, "islogf1", None, fmu, fs2)
    if w_vp < 1:                                                                     # :329-336
        VV = float(np.prod(2 * rect_delta))
        for ii in range(N):
            for a in range(Na):
                inside = bool(np.all(np.abs(Xa[a] - X[ii]) < rect_delta))
                templ[a, ii + 1] = math.log(inside / VV / N * (1 - w_vp)) if inside else -np.inf
        mmax = np.max(templ, axis=1)
        with np.errstate(divide="ignore", invalid="ignore"):      # a point outside every component: exp(-Inf - -Inf) = NaN, as in MATLAB (:338)
            lpdf = np.log(np.sum(np.exp(templ - mmax[:, None]), axis=1))
        lnw = lny - (lpdf + mmax)[:, None]
    else:
        lnw = lny - templ[:, 0:1]
    return lnw, fs2


def acq_iqr(name, Xs, vp, gp, optimState, fmu, fs2, fbar, vtot):
    """acq/acqviqr_vbmc.m:36-109 (name 'acqviqr') and acq/acqimiqr_vbmc.m:30-95 ('acqimiqr'), SE-ARD
    covariance, no integrated mean.  optimState['ActiveImportanceSampling'] holds Xa, fs2a (Na x S), lnw (S x Na)
    and, for VIQR, Ctmp_mat (N x Na x S); IMIQR uses Kax_mat (Na x N x S) and redoes the solve per call."""
    u = ACQ_U
    Xs = np.asarray(Xs, dtype=np.float64)
    Nx, D = Xs.shape
    Ns = fmu.shape[1]
    AIS = optimState["ActiveImportanceSampling"]
    pos = np.argmin(_acq_sqdist_rows(Xs / optimState["gplengthscale"][None, :], gp["X_rescaled"]), axis=1)
    sn2 = np.asarray(gp["sn2new"], dtype=np.float64)[pos]
    ys2 = fs2 + sn2[:, None]  # predictive variance at the test points
    Xa_all = np.asarray(AIS["Xa"], dtype=np.float64)
    acq = np.zeros((Nx, Ns))
    X = gp["X"]
    for s, post in enumerate(gp["post"]):
        hyp = post["hyp"]
        ell = np.exp(hyp[0:D])
        sf2 = math.exp(2.0 * hyp[D])
        Xa = Xa_all if Xa_all.ndim == 2 else Xa_all[:, :, s]
        Xs_ell = Xs / ell[None, :]
        Ks_mat = sf2 * np.exp(-_acq_sqdist_rows(X / ell[None, :], Xs_ell) / 2.0)  # N x Nx
        if name == "acqviqr":
            Ka_mat = sf2 * np.exp(-_acq_sqdist_rows(Xs_ell, Xa / ell[None, :]) / 2.0)  # Nx x Na   (:73-74)
            Ct = AIS["Ctmp_mat"][:, :, s]
            C = Ka_mat - Ks_mat.T @ Ct if post["Lchol"] else Ka_mat + Ks_mat.T @ Ct  # :84-90
            lnw = 0.0  # VIQR: plain Monte Carlo (:103-105)
        else:
            Ka_mat = sf2 * np.exp(-_acq_sqdist_rows(Xa / ell[None, :], Xs_ell) / 2.0)  # Na x Nx   (:67-68)
            Kax = AIS["Kax_mat"][:, :, s]
            L = post["L"]
            if post["Lchol"]:
                sn2_eff = 1.0 / post["sW"][0] ** 2
                C = Ka_mat.T - Ks_mat.T @ (solve_upper(L, solve_upper_t(L, Kax.T))) / sn2_eff  # :77
            else:
                C = Ka_mat.T + Ks_mat.T @ (L @ Kax.T)  # :79
            lnw = np.asarray(AIS["lnw"], dtype=np.float64)[s, :][None, :]  # :85
        tau2 = C**2 / ys2[:, s][:, None]
        s_pred = np.sqrt(np.maximum(np.asarray(AIS["fs2a"])[:, s][None, :] - tau2, 0.0))
        with np.errstate(divide="ignore", invalid="ignore"):
            zz = lnw + u * s_pred + np.log1p(-np.exp(-2 * u * s_pred))
            lnmax = np.max(zz, axis=1)
            acq[:, s] = np.log(np.sum(np.exp(zz - lnmax[:, None]), axis=1)) + lnmax
    if Ns > 1:
        with np.errstate(divide="ignore", invalid="ignore"):
            M = np.max(acq, axis=1)
            return M + np.log(np.sum(np.exp(acq - M[:, None]), axis=1) / Ns)
    return acq[:, 0]


ACQ_LOG_FLAG = {"acqf": False, "acqflog": True, "acqus": False, "acqfsn2": False, "acqviqr": True, "acqimiqr": True}


def acqwrapper_vbmc(Xs, vp, gp, optimState, acq_name, outside=None):
    """acq/acqwrapper_vbmc.m:11-51 for vp.delta = 0, without the integer mapping (:8) and with the hard-bound
    test (:46-49, needs the caller's warpvars inverse) supplied as the boolean mask ``outside``."""
    Xs = np.asarray(Xs, dtype=np.float64)
    _, _, fmu, fs2 = gplite_pred(gp, Xs, None, None, True)  # :17
    fmu = np.asarray(fmu).reshape(Xs.shape[0], -1)
    fs2 = np.asarray(fs2).reshape(Xs.shape[0], -1)
    Ns = fmu.shape[1]
    fbar = np.sum(fmu, axis=1) / Ns  # :22
    vbar = np.sum(fs2, axis=1) / Ns
    vf = np.sum((fmu - fbar[:, None]) ** 2, axis=1) / (Ns - 1) if Ns > 1 else 0.0  # :24-28
    vtot = vf + vbar
    if acq_name in ("acqviqr", "acqimiqr"):
        acq = acq_iqr(acq_name, Xs, vp, gp, optimState, fmu, fs2, fbar, vtot)  # :32
    else:
        acq = acq_function(acq_name, Xs, vp, gp, optimState, fmu, fs2, fbar, vtot)  # :32
    if optimState.get("VarianceRegularizedAcqFcn", False):  # :35-45
        TolVar = optimState["TolGPVar"]
        idx = vtot < TolVar
        if np.any(idx):
            if ACQ_LOG_FLAG[acq_name]:
                acq[idx] = acq[idx] + TolVar / vtot[idx] - 1
            else:
                acq[idx] = acq[idx] * np.exp(-(TolVar / vtot[idx] - 1))
    acq = np.maximum(acq, -REALMAX)  # :46
    if outside is not None:
        acq = np.where(np.asarray(outside, dtype=bool), np.inf, acq)  # :49-51
    return acq, fbar, vtot


def softmax_jacobian(eta):
    """J_w = -exp(eta)' * exp(eta)/sum^2 + diag(exp(eta)/sum)
    (misc/gplogjoint.m:366-368, ent/entmc_vbmc.m:121-123)."""
    e = np.exp(np.asarray(eta, dtype=np.float64))
    eta_sum = np.sum(e)
    return -np.outer(e, e / eta_sum**2) + np.diag(e / eta_sum)


# --------------------------------------------------------------------------
# gplogjoint
# --------------------------------------------------------------------------


def _unpack_hyp(gp, hyp, D):
    """misc/gplogjoint.m:99-121 for meanfun in {0,1,4}."""
    Ncov, Nnoise = gp["Ncov"], gp["Nnoise"]
    ell = np.exp(hyp[0:D])
    ln_sf2 = 2.0 * hyp[D]
    sum_lnell = np.sum(hyp[0:D])
    meanfun = gp["meanfun"]
    m0 = hyp[Ncov + Nnoise] if meanfun > 0 else 0.0  # :107-111
    xm = omega = None
    if meanfun == 4:  # :112-121 (quadratic_meanfun, not fixed)
        xm = hyp[Ncov + Nnoise + 1 : Ncov + Nnoise + 1 + D]
        omega = np.exp(hyp[Ncov + Nnoise + D + 1 : Ncov + Nnoise + 2 * D + 1])
    return ell, ln_sf2, sum_lnell, m0, xm, omega


def gplogjoint(vp, gp, grad_flags=(0, 0, 0, 0), avg_flag=True, jacobian_flag=True, compute_var=0,
               separate_K=False, compute_vargrad=None):
    """misc/gplogjoint.m:1-415 -> dict(F, dF, varF, dvarF, varss, I_sk, J_sjk).

    compute_var: 0 none, 1 full K x K (pair loop :306-337), 2 diagonal (:273-304).
    ``compute_vargrad`` mirrors ``nargout > 3 && compute_var && any(grad_flags)``
    (:27); pass True to request dvarF (requires compute_var == 2).
    """
    if gp["meanfun"] not in (0, 1, 4):
        raise NotImplementedError("gplogjoint:UnsupportedMeanFun (only 0,1,4 restated)")
    grad_flags = tuple(bool(g) for g in (grad_flags if np.ndim(grad_flags) else (grad_flags,) * 4))
    anygrad = any(grad_flags)
    if compute_vargrad is None:
        compute_vargrad = False
    compute_vargrad = bool(compute_vargrad and compute_var and anygrad)
    if compute_vargrad and compute_var != 2:
        raise ValueError("gplogjoint:FullVarianceGradient")  # :29-32

    D, K = vp["D"], vp["K"]
    X = gp["X"]
    N = X.shape[0]
    mu, sigma, lam, w = vp["mu"], vp["sigma"], vp["lambda"], vp["w"]
    S = len(gp["post"])
    quadratic = gp["meanfun"] == 4

    F = np.zeros(S)
    mu_grad = np.zeros((D, K, S)) if grad_flags[0] else None
    sigma_grad = np.zeros((K, S)) if grad_flags[1] else None
    lambda_grad = np.zeros((D, S)) if grad_flags[2] else None
    w_grad = np.zeros((K, S)) if grad_flags[3] else None
    varF = np.zeros(S) if compute_var else None
    if compute_vargrad:
        mu_vargrad = np.zeros((D, K, S)) if grad_flags[0] else None
        sigma_vargrad = np.zeros((K, S)) if grad_flags[1] else None
        lambda_vargrad = np.zeros((D, S)) if grad_flags[2] else None
        w_vargrad = np.zeros((K, S)) if grad_flags[3] else None
    I_sk = np.zeros((S, K)) if separate_K else None
    J_sjk = np.zeros((S, K, K)) if (separate_K and compute_var) else None

    delta = vp.get("delta")
    if delta is None or np.size(delta) == 0:
        delta = 0.0  # :85-89
    delta = np.asarray(delta, dtype=np.float64).reshape(-1) if np.ndim(delta) else float(delta)

    Xt = mu.T[:, :, None] - X.T[None, :, :]  # Xt[k] = mu(:,k) - X'  (D x N)  :92-95

    for s in range(S):
        post = gp["post"][s]
        hyp = post["hyp"]
        ell, ln_sf2, sum_lnell, m0, xm, omega = _unpack_hyp(gp, hyp, D)
        alpha = post["alpha"]
        L = post["L"]
        Lchol = post["Lchol"]
        sn2_eff = 1.0 / post["sW"][0] ** 2  # :160

        for k in range(K):
            tau_k = np.sqrt(sigma[k] ** 2 * lam**2 + ell**2 + delta**2)  # :164
            lnnf_k = ln_sf2 + sum_lnell - np.sum(np.log(tau_k))  # :165
            delta_k = Xt[k] / tau_k[:, None]  # :167
            z_k = np.exp(lnnf_k - 0.5 * np.sum(delta_k**2, axis=0))  # :168
            I_k = z_k @ alpha + m0  # :169
            if quadratic:
                nu_k = -0.5 * np.sum(
                    1.0 / omega**2 * (mu[:, k] ** 2 + sigma[k] ** 2 * lam**2 - 2.0 * mu[:, k] * xm + xm**2 + delta**2)
                )  # :172-173
                I_k = I_k + nu_k
            F[s] += w[k] * I_k  # :203
            if separate_K:
                I_sk[s, k] = I_k

            if grad_flags[0]:
                dz_dmu = -(delta_k / tau_k[:, None]) * z_k[None, :]  # :207
                mu_grad[:, k, s] = w[k] * (dz_dmu @ alpha)  # :208
                if quadratic:
                    mu_grad[:, k, s] -= w[k] / omega**2 * (mu[:, k] - xm)  # :210
            if grad_flags[1]:
                dz_dsigma = np.sum((lam / tau_k)[:, None] ** 2 * (delta_k**2 - 1.0), axis=0) * (sigma[k] * z_k)  # :228
                sigma_grad[k, s] = w[k] * (dz_dsigma @ alpha)  # :229
                if quadratic:
                    sigma_grad[k, s] -= w[k] * sigma[k] * np.sum(1.0 / omega**2 * lam**2)  # :231
            if grad_flags[2]:
                dz_dlambda = ((sigma[k] / tau_k) ** 2)[:, None] * (delta_k**2 - 1.0) * (lam[:, None] * z_k[None, :])  # :249
                lambda_grad[:, s] += w[k] * (dz_dlambda @ alpha)  # :250
                if quadratic:
                    lambda_grad[:, s] -= w[k] * sigma[k] ** 2 / omega**2 * lam  # :252
            if grad_flags[3]:
                w_grad[k, s] = I_k  # :270

            if compute_var == 2:  # :273-304
                tau_kk = np.sqrt(2.0 * sigma[k] ** 2 * lam**2 + ell**2 + 2.0 * delta**2)
                nf_kk = math.exp(ln_sf2 + sum_lnell - np.sum(np.log(tau_kk)))
                if Lchol:
                    invKzk = solve_upper(L, solve_upper_t(L, z_k)) / sn2_eff  # :277
                else:
                    invKzk = -L @ z_k  # :279
                J_kk = nf_kk - z_k @ invKzk  # :281
                varF[s] += w[k] ** 2 * max(EPS, J_kk)  # :283
                if separate_K:
                    J_sjk[s, k, k] = J_kk
                if compute_vargrad:
                    if grad_flags[0]:
                        mu_vargrad[:, k, s] = -w[k] ** 2 * (2.0 * (dz_dmu @ invKzk))  # :289
                    if grad_flags[1]:
                        sigma_vargrad[k, s] = -2.0 * w[k] ** 2 * (
                            sigma[k] * nf_kk * np.sum(lam**2 / tau_kk**2) + dz_dsigma @ invKzk
                        )  # :293
                    if grad_flags[2]:
                        lambda_vargrad[:, s] -= 2.0 * w[k] ** 2 * (
                            sigma[k] ** 2 * nf_kk * lam / tau_kk**2 + dz_dlambda @ invKzk
                        )  # :297
                    if grad_flags[3]:
                        w_vargrad[k, s] = 2.0 * w[k] * max(EPS, J_kk)  # :301
            elif compute_var:  # :306-337  full pair loop
                for j in range(k + 1):
                    tau_j = np.sqrt(sigma[j] ** 2 * lam**2 + ell**2 + delta**2)
                    lnnf_j = ln_sf2 + sum_lnell - np.sum(np.log(tau_j))
                    delta_j = (mu[:, j][:, None] - X.T) / tau_j[:, None]
                    z_j = np.exp(lnnf_j - 0.5 * np.sum(delta_j**2, axis=0))
                    tau_jk = np.sqrt((sigma[j] ** 2 + sigma[k] ** 2) * lam**2 + ell**2 + 2.0 * delta**2)
                    lnnf_jk = ln_sf2 + sum_lnell - np.sum(np.log(tau_jk))
                    delta_jk = (mu[:, j] - mu[:, k]) / tau_jk
                    if Lchol:
                        J_jk = math.exp(lnnf_jk - 0.5 * np.sum(delta_jk**2)) - z_k @ solve_upper(
                            L, solve_upper_t(L, z_j)
                        ) / sn2_eff  # :318-319
                    else:
                        J_jk = math.exp(lnnf_jk - 0.5 * np.sum(delta_jk**2)) + z_k @ (L @ z_j)  # :321-322
                    if j == k:
                        varF[s] += w[k] ** 2 * max(EPS, J_jk)  # :329
                        if separate_K:
                            J_sjk[s, k, k] = J_jk
                    else:
                        varF[s] += 2.0 * w[j] * w[k] * J_jk  # :332
                        if separate_K:
                            J_sjk[s, j, k] = J_jk
                            J_sjk[s, k, j] = J_jk

    if compute_var:
        varF = np.maximum(varF, EPS)  # :350

    dF = None
    J_w = None
    if anygrad:  # :352-373
        blocks = []
        if grad_flags[0]:
            blocks.append(mu_grad.reshape(D * K, S, order="F"))
        if grad_flags[1]:
            if jacobian_flag:
                sigma_grad = sigma_grad * sigma[:, None]
            blocks.append(sigma_grad)
        if grad_flags[2]:
            if jacobian_flag:
                lambda_grad = lambda_grad * lam[:, None]
            blocks.append(lambda_grad)
        if grad_flags[3]:
            if jacobian_flag:
                J_w = softmax_jacobian(vp["eta"])
                w_grad = J_w @ w_grad
            blocks.append(w_grad)
        dF = np.vstack(blocks)

    dvarF = None
    if compute_vargrad:  # :375-395
        blocks = []
        if grad_flags[0]:
            blocks.append(mu_vargrad.reshape(D * K, S, order="F"))
        if grad_flags[1]:
            if jacobian_flag:
                sigma_vargrad = sigma_vargrad * sigma[:, None]
            blocks.append(sigma_vargrad)
        if grad_flags[2]:
            if jacobian_flag:
                lambda_vargrad = lambda_vargrad * lam[:, None]
            blocks.append(lambda_vargrad)
        if grad_flags[3]:
            if jacobian_flag:
                w_vargrad = J_w @ w_vargrad
            blocks.append(w_vargrad)
        dvarF = np.vstack(blocks)

    varss = 0.0
    if S > 1 and avg_flag:  # :399-413
        Fbar = np.sum(F) / S
        if compute_var:
            varFss = np.sum((F - Fbar) ** 2) / (S - 1)
            varss = varFss + np.std(varF, ddof=1)  # MATLAB std normalises by S-1
            varF_out = np.sum(varF) / S + varFss
        if compute_vargrad:
            dvv = 2.0 * np.sum(F[None, :] * dF, axis=1) / (S - 1) - 2.0 * Fbar * np.sum(dF, axis=1) / (S - 1)
            dvarF = np.sum(dvarF, axis=1) / S + dvv
        if compute_var:
            varF = varF_out
        F = Fbar
        if anygrad:
            dF = np.sum(dF, axis=1) / S
    else:
        if S == 1:  # MATLAB 1x1 / Tx1 results
            F = F[0]
            if compute_var:
                varF = varF[0]
            if anygrad:
                dF = dF[:, 0]
            if compute_vargrad:
                dvarF = dvarF[:, 0]
    return {"F": F, "dF": dF, "varF": varF, "dvarF": dvarF, "varss": varss, "I_sk": I_sk, "J_sjk": J_sjk}


def gplogjoint_weights(vp, grad_flag, avg_flag=True, jacobian_flag=True, compute_var=0, compute_vargrad=False):
    """misc/gplogjoint_weights.m:1-104 (reuses cached vp.stats.I_sk / J_sjk)."""
    K = vp["K"]
    w = vp["w"]
    I_sk = vp["stats"]["I_sk"]
    J_sjk = vp["stats"]["J_sjk"]
    S = I_sk.shape[0]
    compute_vargrad = bool(compute_vargrad and compute_var and grad_flag)
    if compute_vargrad and compute_var != 2:
        raise ValueError("gplogjoint:FullVarianceGradient")
    F = np.zeros(S)
    w_grad = np.zeros((K, S)) if grad_flag else None
    varF = np.zeros(S) if compute_var else None
    w_vargrad = np.zeros((K, S)) if compute_vargrad else None
    for s in range(S):
        F[s] = np.sum(w * I_sk[s, :])  # :47
        if grad_flag:
            w_grad[:, s] = I_sk[s, :]
        if compute_var == 2:
            J_diag = np.diag(J_sjk[s])
            varF[s] = np.sum(w**2 * np.maximum(EPS, J_diag))  # :52
            if compute_vargrad:
                w_vargrad[:, s] = 2.0 * w * np.maximum(EPS, J_diag)
        elif compute_var:
            varF[s] = np.sum(J_sjk[s] * np.outer(w, w))  # :58
    if compute_var:
        varF = np.maximum(varF, EPS)
    dF = None
    J_w = None
    if grad_flag:
        if jacobian_flag:
            J_w = softmax_jacobian(vp["eta"])
            w_grad = J_w @ w_grad
        dF = w_grad
    dvarF = None
    if compute_vargrad:
        if jacobian_flag:
            w_vargrad = J_w @ w_vargrad
        dvarF = w_vargrad
    varss = 0.0
    if S > 1 and avg_flag:
        Fbar = np.sum(F) / S
        if compute_var:
            varFss = np.sum((F - Fbar) ** 2) / (S - 1)
            varss = varFss + np.std(varF, ddof=1)
            varF_out = np.sum(varF) / S + varFss
        if compute_vargrad:
            dvv = 2.0 * np.sum(F[None, :] * dF, axis=1) / (S - 1) - 2.0 * Fbar * np.sum(dF, axis=1) / (S - 1)
            dvarF = np.sum(dvarF, axis=1) / S + dvv
        if compute_var:
            varF = varF_out
        F = Fbar
        if grad_flag:
            dF = np.sum(dF, axis=1) / S
    elif S == 1:
        F = F[0]
        if compute_var:
            varF = varF[0]
        if grad_flag:
            dF = dF[:, 0]
        if compute_vargrad:
            dvarF = dvarF[:, 0]
    return {"F": F, "dF": dF, "varF": varF, "dvarF": dvarF, "varss": varss, "I_sk": I_sk, "J_sjk": J_sjk}


# --------------------------------------------------------------------------
# entmc_vbmc
# --------------------------------------------------------------------------


def entmc_vbmc(vp, Ns, grad_flags=(0, 0, 0, 0), jacobian_flag=True, eps=None, rng=None):
    """ent/entmc_vbmc.m:1-128 -> (H, dH).

    ``eps``: (K, ceil(Ns/2), D) standard normals standing in for the K calls
    ``randn(D,1,Ns/2)`` at :53; drawn from ``rng`` (np.random.Generator) if None.
    """
    grad_flags = tuple(bool(g) for g in (grad_flags if np.ndim(grad_flags) else (grad_flags,) * 4))
    D, K = vp["D"], vp["K"]
    mu, sigma, lam, w = vp["mu"], vp["sigma"], vp["lambda"], vp["w"]
    mu_grad = np.zeros((D, K)) if grad_flags[0] else None
    sigma_grad = np.zeros(K) if grad_flags[1] else None
    lambda_grad = np.zeros(D) if grad_flags[2] else None
    w_grad = np.zeros(K) if grad_flags[3] else None

    sigmalambda = sigma[None, :] * lam[:, None]  # D x K   :34
    nconst = 1.0 / (2.0 * math.pi) ** (D / 2.0) / np.prod(lam)  # :35
    nf = nconst  # :40
    H = 0.0
    Ns = int(math.ceil(Ns / 2.0) * 2)  # :45
    Mh = Ns // 2
    if eps is None:
        rng = np.random.default_rng() if rng is None else rng
        eps = rng.standard_normal((K, Mh, D))
    eps = np.asarray(eps, dtype=np.float64)
    assert eps.shape == (K, Mh, D), (eps.shape, (K, Mh, D))

    for j in range(K):
        epsilon = np.concatenate([eps[j], -eps[j]], axis=0)  # Ns x D  :53-54 antithetic
        xi = epsilon * lam[None, :] * sigma[j] + mu[:, j][None, :]  # Ns x D  :55
        Xs = xi
        ys = np.zeros(Ns)
        for k in range(K):  # :60-65
            d2 = np.sum(((Xs - mu[:, k][None, :]) / (sigma[k] * lam[None, :])) ** 2, axis=1)
            nn = w[k] * nf / sigma[k] ** D * np.exp(-0.5 * d2)
            ys = ys + nn
        H = H - w[j] * np.sum(np.log(ys)) / Ns  # :67

        if any(grad_flags):
            diff = xi[:, :, None] - mu[None, :, :]  # Ns x D x K
            norm_jl = (nconst / sigma**D)[None, :] * np.exp(-0.5 * np.sum((diff / sigmalambda[None, :, :]) ** 2, axis=1))  # Ns x K  :72
            q_j = np.sum(w[None, :] * norm_jl, axis=1)  # Ns  :73
            lsum = np.sum(diff / sigmalambda[None, :, :] ** 2 * (norm_jl * w[None, :])[:, None, :], axis=2)  # Ns x D :77-79
            if grad_flags[0]:
                mu_grad[:, j] = w[j] * np.sum(lsum / q_j[:, None], axis=0) / Ns  # :82
            if grad_flags[1]:
                isum = np.sum(lsum * (epsilon * lam[None, :]), axis=1)  # :87
                sigma_grad[j] = w[j] * np.sum(isum / q_j) / Ns  # :88
            if grad_flags[2]:
                lambda_grad = lambda_grad + np.sum(lsum * (w[j] * sigma[j] * epsilon / q_j[:, None]), axis=0) / Ns  # :93
            if grad_flags[3]:
                w_grad[j] = w_grad[j] - np.sum(np.log(q_j)) / Ns  # :97
                w_grad = w_grad - w[j] * np.sum(norm_jl / q_j[:, None], axis=0) / Ns  # :100

    if grad_flags[2]:
        lambda_grad = lambda_grad * lam  # :107
    if jacobian_flag and grad_flags[1]:
        sigma_grad = sigma_grad * sigma  # :113
    if (not jacobian_flag) and grad_flags[2]:
        lambda_grad = lambda_grad / lam  # :117
    if jacobian_flag and grad_flags[3]:
        w_grad = softmax_jacobian(vp["eta"]) @ w_grad  # :121-123
    blocks = []
    if grad_flags[0]:
        blocks.append(mu_grad.reshape(-1, order="F"))
    if grad_flags[1]:
        blocks.append(sigma_grad)
    if grad_flags[2]:
        blocks.append(lambda_grad)
    if grad_flags[3]:
        blocks.append(w_grad)
    dH = np.concatenate(blocks) if blocks else np.zeros(0)
    return H, dH


# --------------------------------------------------------------------------
# entlb_vbmc
# --------------------------------------------------------------------------


def entlb_vbmc(vp, grad_flags=(0, 0, 0, 0), jacobian_flag=True):
    """ent/entlb_vbmc.m:1-148 (Gershman et al. lower bound) -> (H, dH)."""
    grad_flags = tuple(bool(g) for g in (grad_flags if np.ndim(grad_flags) else (grad_flags,) * 4))
    D, K = vp["D"], vp["K"]
    mu, sigma, lam, w = vp["mu"], vp["sigma"], vp["lambda"], vp["w"]
    mu_grad = np.zeros((D, K)) if grad_flags[0] else None
    sigma_grad = np.zeros(K) if grad_flags[1] else None
    lambda_grad = np.zeros(D) if grad_flags[2] else None
    w_grad = np.zeros(K) if grad_flags[3] else None

    if K == 1:  # :32-47
        H = 0.5 * D * (1.0 + math.log(2.0 * math.pi)) + D * np.sum(np.log(sigma)) + np.sum(np.log(lam))
        if grad_flags[1]:
            sigma_grad[:] = D / sigma
        if grad_flags[2]:
            lambda_grad[:] = 1.0
        if grad_flags[3]:
            w_grad = np.zeros(1)
    else:  # :66-126
        sumsigma2 = sigma[:, None] ** 2 + sigma[None, :] ** 2  # [j, k]
        sumsigma = np.sqrt(sumsigma2)
        nconst = 1.0 / (2.0 * math.pi) ** (D / 2.0) / np.prod(lam)
        dmu_jk = mu[:, :, None] - mu[:, None, :]  # D x j x k : mu_j - mu_k
        d2 = np.sum((dmu_jk / (sumsigma[None, :, :] * lam[:, None, None])) ** 2, axis=0)  # :74
        gamma = nconst / sumsigma**D * np.exp(-0.5 * d2)  # [j,k]  :75
        gammasum = np.sum(w[:, None] * gamma, axis=0)  # over j -> [k]  :76
        H = -np.sum(w * np.log(gammasum))  # :78
        if any(grad_flags):
            gammafrac = gamma / gammasum[None, :]  # :83
            wgammafrac = w[None, :] * gammafrac  # w_3 (3rd dim = k) :84
            if grad_flags[0]:
                dmu = (mu[:, None, :] - mu[:, :, None]) / (sumsigma2[None, :, :] * lam[:, None, None] ** 2)  # mu_k - mu_j  :87
            if grad_flags[1]:
                dsigma = -D / sumsigma2 + 1.0 / sumsigma2**2 * np.sum((dmu_jk / lam[:, None, None]) ** 2, axis=0)  # :90
            for j in range(K):
                if grad_flags[0]:
                    m1 = np.sum(wgammafrac[j, :][None, :] * dmu[:, j, :], axis=1)  # :96
                    m2 = np.sum(dmu[:, j, :] * (gamma[j, :] * w)[None, :], axis=1) / gammasum[j]  # :97
                    mu_grad[:, j] = -w[j] * (m1 + m2)  # :98
                if grad_flags[1]:
                    s1 = np.sum(wgammafrac[j, :] * dsigma[j, :])  # :103
                    s2 = np.sum(dsigma[j, :] * gamma[j, :] * w) / gammasum[j]  # :104
                    sigma_grad[j] = -w[j] * sigma[j] * (s1 + s2)  # :105
            if grad_flags[2]:
                dmu2 = (mu[:, None, :] - mu[:, :, None]) ** 2 / (sumsigma2[None, :, :] * lam[:, None, None] ** 2)  # :110
                inner = np.sum((dmu2 - 1.0) * (gamma * w[:, None])[None, :, :], axis=1)  # sum over j -> D x k :112
                lambda_grad[:] = -np.sum(w[None, :] * inner / gammasum[None, :], axis=1)  # :111-113
            if grad_flags[3]:
                w_grad[:] = -np.log(gammasum) - np.sum(wgammafrac, axis=1)  # :118

    if jacobian_flag and grad_flags[1]:
        sigma_grad = sigma_grad * sigma  # :131
    if (not jacobian_flag) and grad_flags[2]:
        lambda_grad = lambda_grad / lam  # :135
    if jacobian_flag and grad_flags[3]:
        w_grad = softmax_jacobian(vp["eta"]) @ w_grad  # :139-141
    blocks = []
    if grad_flags[0]:
        blocks.append(mu_grad.reshape(-1, order="F"))
    if grad_flags[1]:
        blocks.append(sigma_grad)
    if grad_flags[2]:
        blocks.append(lambda_grad)
    if grad_flags[3]:
        blocks.append(np.asarray(w_grad).reshape(-1))
    dH = np.concatenate(blocks) if blocks else np.zeros(0)
    return H, dH


# --------------------------------------------------------------------------
# soft bounds
# --------------------------------------------------------------------------


def softbndloss(x, slb, sub, TolCon=1e-3):
    """utils/softbndloss.m:1-30 -> (y, dy)."""
    x = np.asarray(x, dtype=np.float64)
    ell = (sub - slb) * TolCon
    y = 0.0
    dy = np.zeros_like(x)
    idx = x < slb
    if np.any(idx):
        y += 0.5 * np.sum(((slb[idx] - x[idx]) / ell[idx]) ** 2)
        dy[idx] = (x[idx] - slb[idx]) / ell[idx] ** 2
    idx = x > sub
    if np.any(idx):
        y += 0.5 * np.sum(((x[idx] - sub[idx]) / ell[idx]) ** 2)
        dy[idx] = (x[idx] - sub[idx]) / ell[idx] ** 2
    return y, dy


def vpbounds(vp, gp, options, K=None):
    """misc/vpbounds.m:1-55 -> (vp, thetabnd); accumulates into vp['bounds']."""
    if K is None:
        K = vp["K"]
    D = vp["D"]
    vp = copy_vp(vp)
    b = vp.get("bounds")
    if not b:
        b = {
            "mu_lb": np.full(D, np.inf),
            "mu_ub": np.full(D, -np.inf),
            "lnscale_lb": np.full(D, np.inf),
            "lnscale_ub": np.full(D, -np.inf),
        }
    X = gp["X"]
    b["mu_lb"] = np.minimum(X.min(axis=0), b["mu_lb"])  # :18
    b["mu_ub"] = np.maximum(X.max(axis=0), b["mu_ub"])
    lnrange = np.log(X.max(axis=0) - X.min(axis=0))  # :22
    b["lnscale_lb"] = np.minimum(b["lnscale_lb"], lnrange + math.log(options["TolLength"]))
    b["lnscale_ub"] = np.maximum(b["lnscale_ub"], lnrange)
    if vp["optimize_weights"]:
        b["eta_lb"] = math.log(0.5 * options["TolWeight"])  # :28
        b["eta_ub"] = 0.0
    vp["bounds"] = b
    lb, ub = [], []
    if vp["optimize_mu"]:
        lb.append(np.tile(b["mu_lb"], K))
        ub.append(np.tile(b["mu_ub"], K))
    if vp["optimize_sigma"] or vp["optimize_lambda"]:
        lb.append(np.tile(b["lnscale_lb"], K))
        ub.append(np.tile(b["lnscale_ub"], K))
    if vp["optimize_weights"]:
        lb.append(np.full(K, b["eta_lb"]))
        ub.append(np.full(K, b["eta_ub"]))
    thetabnd = {
        "lb": np.concatenate(lb) if lb else np.zeros(0),
        "ub": np.concatenate(ub) if ub else np.zeros(0),
        "TolCon": options["TolConLoss"],
    }
    if vp["optimize_weights"]:
        thetabnd["WeightThreshold"] = max(1.0 / (4 * K), options["TolWeight"])  # :51
        thetabnd["WeightPenalty"] = options["WeightPenalty"]
    return vp, thetabnd


def vpbndloss(theta, vp, thetabnd, TolCon, compute_grad=True):
    """misc/vpbndloss.m:1-73 -> (L, dL)."""
    theta = np.asarray(theta, dtype=np.float64).reshape(-1)
    K, D = vp["K"], vp["D"]
    if vp["optimize_mu"]:
        mu = theta[0 : K * D]
        idx_start = K * D
    else:
        mu = vp["mu"].reshape(-1, order="F")
        idx_start = 0
    if vp["optimize_sigma"]:
        lnsigma = theta[idx_start : idx_start + K]
        idx_start += K
    else:
        lnsigma = np.log(vp["sigma"])
    if vp["optimize_lambda"]:
        lnlambda = theta[idx_start : idx_start + D]
    else:
        lnlambda = np.log(vp["lambda"])
    eta = theta[-K:] if vp["optimize_weights"] else None
    lnscale = lnsigma[None, :] + lnlambda[:, None]  # D x K  :36
    ext = []
    if vp["optimize_mu"]:
        ext.append(mu)
    if vp["optimize_sigma"] or vp["optimize_lambda"]:
        ext.append(lnscale.reshape(-1, order="F"))
    if vp["optimize_weights"]:
        ext.append(eta)
    theta_ext = np.concatenate(ext)
    L, dLext = softbndloss(theta_ext, thetabnd["lb"], thetabnd["ub"], TolCon)
    if not compute_grad:
        return L, None
    out = []
    if vp["optimize_mu"]:
        out.append(dLext[0 : D * K])
        idx_start = D * K
    else:
        idx_start = 0
    if vp["optimize_sigma"] or vp["optimize_lambda"]:
        dlnscale = dLext[idx_start : idx_start + D * K].reshape(D, K, order="F")
        if vp["optimize_sigma"]:
            out.append(np.sum(dlnscale, axis=0))
        if vp["optimize_lambda"]:
            out.append(np.sum(dlnscale, axis=1))
    if vp["optimize_weights"]:
        out.append(dLext[-K:])
    return L, np.concatenate(out)


# --------------------------------------------------------------------------
# theta <-> vp
# --------------------------------------------------------------------------


def rescale_params(vp, theta=None):
    """misc/rescale_params.m:1-40."""
    vp = copy_vp(vp)
    D = vp["D"]
    if theta is not None and np.size(theta) > 0:
        theta = np.asarray(theta, dtype=np.float64).reshape(-1)
        K = vp["K"]
        if vp["optimize_mu"]:
            vp["mu"] = theta[0 : D * K].reshape(D, K, order="F").copy()
            idx_start = D * K
        else:
            idx_start = 0
        if vp["optimize_sigma"]:
            vp["sigma"] = np.exp(theta[idx_start : idx_start + K])
            idx_start += K
        if vp["optimize_lambda"]:
            vp["lambda"] = np.exp(theta[idx_start : idx_start + D])
        if vp["optimize_weights"]:
            eta = theta[-K:]
            eta = eta - np.max(eta)  # :23
            vp["w"] = np.exp(eta)
    nl = math.sqrt(np.sum(vp["lambda"] ** 2) / D)  # :28
    vp["lambda"] = vp["lambda"] / nl
    vp["sigma"] = vp["sigma"] * nl
    if vp["optimize_weights"]:
        vp["w"] = vp["w"] / np.sum(vp["w"])  # :34
        vp.pop("eta", None)
    vp.pop("mode", None)
    return vp


def get_vptheta(vp, optimize_mu=None, optimize_sigma=None, optimize_lambda=None, optimize_weights=None):
    """misc/get_vptheta.m:1-22 -> (theta, vp)."""
    if optimize_weights is None:
        optimize_weights = vp["optimize_weights"]
    if optimize_lambda is None:
        optimize_lambda = vp["optimize_lambda"]
    if optimize_sigma is None:
        optimize_sigma = vp["optimize_sigma"]
    if optimize_mu is None:
        optimize_mu = vp["optimize_mu"]
    vp = rescale_params(vp)
    parts = []
    if optimize_mu:
        parts.append(vp["mu"].reshape(-1, order="F"))
    if optimize_sigma:
        parts.append(np.log(vp["sigma"]))
    if optimize_lambda:
        parts.append(np.log(vp["lambda"]))
    if optimize_weights:
        parts.append(np.log(vp["w"]))
    theta = np.concatenate(parts) if parts else np.zeros(0)
    return theta, vp


# --------------------------------------------------------------------------
# negelcbo_vbmc
# --------------------------------------------------------------------------


def negelcbo_vbmc(theta, beta, vp, gp, Ns=0, compute_grad=True, compute_var=None, thetabnd=None,
                  separate_K=False, eps=None, rng=None):
    """misc/negelcbo_vbmc.m:1-165.

    Returns dict(F, dF, G, H, varF, dH, varGss, varG, varH, I_sk, J_sjk).
    ``separate_K`` mirrors ``nargout > 9`` (:17); ``compute_var`` default mirrors
    ``beta ~= 0 || nargout > 4`` only in its first clause (callers pass it).
    """
    theta = np.asarray(theta, dtype=np.float64).reshape(-1)
    if beta is None or not np.isfinite(beta):
        beta = 0.0
    if compute_var is None:
        compute_var = 1 if beta != 0 else 0
    compute_var = int(compute_var)
    if compute_grad and beta != 0 and compute_var != 2:
        raise ValueError("negelcbo_vbmc:vargrad")  # :21-24
    vp = copy_vp(vp)
    D, K = vp["D"], vp["K"]
    # :33-48
    if vp["optimize_mu"]:
        vp["mu"] = theta[0 : D * K].reshape(D, K, order="F").copy()
        idx_start = D * K
    else:
        idx_start = 0
    if vp["optimize_sigma"]:
        vp["sigma"] = np.exp(theta[idx_start : idx_start + K])
        idx_start += K
    if vp["optimize_lambda"]:
        vp["lambda"] = np.exp(theta[idx_start : idx_start + D])
    if vp["optimize_weights"]:
        vp["eta"] = theta[-K:].copy()
        vp["w"] = np.exp(vp["eta"])
        vp["w"] = vp["w"] / np.sum(vp["w"])  # no max-shift (:45-47)

    gf = (vp["optimize_mu"], vp["optimize_sigma"], vp["optimize_lambda"], vp["optimize_weights"])
    grad_flags = tuple(bool(compute_grad) and bool(g) for g in gf)  # :51
    onlyweights = vp["optimize_weights"] and not (vp["optimize_mu"] or vp["optimize_sigma"] or vp["optimize_lambda"])  # :54
    if separate_K and compute_grad:
        raise ValueError("gradient and per-component results requested together")  # :57-59

    dvarG = None
    if onlyweights:
        r = gplogjoint_weights(vp, bool(compute_grad), True, True, compute_var,
                               compute_vargrad=bool(compute_grad and compute_var))
        varGss = float("nan")
    else:
        r = gplogjoint(vp, gp, grad_flags, True, True, compute_var, separate_K=separate_K,
                       compute_vargrad=bool(compute_grad and compute_var))
        varGss = r["varss"] if compute_var else 0.0
    G = r["F"]
    dG = r["dF"]
    varG = r["varF"] if compute_var else 0.0
    dvarG = r["dvarF"]
    I_sk = r["I_sk"] if separate_K else None
    J_sjk = r["J_sjk"] if (separate_K and compute_var) else None

    if Ns > 0:
        H, dH = entmc_vbmc(vp, Ns, grad_flags, True, eps=eps, rng=rng)  # :106
    else:
        H, dH = entlb_vbmc(vp, grad_flags, True)  # :109

    F = -G - H  # :116
    dF = (-dG - dH) if compute_grad else None
    varH = 0.0
    varF = (varG + varH) if compute_var else 0.0
    if beta != 0:
        F = F + beta * math.sqrt(varF)  # :127
        if compute_grad:
            dF = dF + 0.5 * beta * dvarG / math.sqrt(varF)  # :129

    if thetabnd is not None:  # :136-164
        L, dL = vpbndloss(theta, vp, thetabnd, thetabnd["TolCon"], compute_grad=bool(compute_grad))
        if compute_grad:
            dF = dF + dL
        F = F + L
        if vp["optimize_weights"]:
            Thresh = thetabnd["WeightThreshold"]
            w = vp["w"]
            L = np.sum(w * (w < Thresh) + Thresh * (w >= Thresh)) * thetabnd["WeightPenalty"]  # :150
            F = F + L
            if compute_grad:
                w_grad = thetabnd["WeightPenalty"] * (w < Thresh).astype(np.float64)  # :155
                w_grad = softmax_jacobian(vp["eta"]) @ w_grad
                dL = np.zeros_like(dF)
                dL[-K:] = w_grad
                dF = dF + dL
    return {"F": F, "dF": dF, "G": G, "H": H, "varF": varF, "dH": dH if compute_grad else None,
            "dG": dG if compute_grad else None,
            "varGss": varGss, "varG": varG, "varH": varH, "I_sk": I_sk, "J_sjk": J_sjk}


# --------------------------------------------------------------------------
# fminadam
# --------------------------------------------------------------------------


def fminadam(fun, x0, TolFun=0.001, MaxIter=10000, master_stepsize=None, LB=None, UB=None):
    """utils/fminadam.m:1-104 -> (x, f, xtab, ftab, iter)."""
    ms = {"max": 0.1, "min": 0.001, "decay": 200.0}
    if master_stepsize:
        ms.update({k: v for k, v in master_stepsize.items() if v is not None})
    fudge_factor = math.sqrt(EPS)
    beta1, beta2 = 0.9, 0.999
    batchsize = 20
    TolX, TolX_max = 0.001, 0.1
    TolFun_max = TolFun * 100.0
    MinIter = batchsize * 2
    x0 = np.asarray(x0, dtype=np.float64)
    nvars = x0.size
    LB = np.full(nvars, -np.inf) if LB is None else np.asarray(LB, dtype=np.float64).reshape(-1)
    UB = np.full(nvars, np.inf) if UB is None else np.asarray(UB, dtype=np.float64).reshape(-1)
    m = np.zeros(nvars)
    v = np.zeros(nvars)
    xtab = np.zeros((nvars, MaxIter))
    x = x0.reshape(-1).copy()
    ftab = np.full(MaxIter, np.nan)
    it = 0
    for it in range(1, MaxIter + 1):
        isMinibatchEnd = (it % batchsize) == 0
        f, grad = fun(x)  # :48
        ftab[it - 1] = f
        grad = np.asarray(grad, dtype=np.float64).reshape(-1)
        m = beta1 * m + (1 - beta1) * grad
        v = beta2 * v + (1 - beta2) * grad**2
        mhat = m / (1 - beta1**it)
        vhat = v / (1 - beta2**it)
        stepsize = ms["min"] + (ms["max"] - ms["min"]) * math.exp(-it / ms["decay"])  # :56-57
        x = x - stepsize * mhat / (np.sqrt(vhat) + fudge_factor)  # :59
        x = np.minimum(np.maximum(x, LB), UB)
        xtab[:, it - 1] = x
        if isMinibatchEnd and it >= MinIter:  # :65-81
            xxp = np.linspace(-(batchsize - 1) / 2.0, (batchsize - 1) / 2.0, batchsize)
            yy = ftab[it - batchsize : it]
            slope, slope_var = _polyfit1_slope(xxp, yy)
            slope_err = math.sqrt(slope_var + TolFun**2)
            slope_err_max = math.sqrt(slope_var + TolFun_max**2)
            dx = math.sqrt(
                np.sum(
                    (np.mean(xtab[:, it - batchsize : it], axis=1) - np.mean(xtab[:, it - 2 * batchsize : it - batchsize], axis=1)) ** 2
                    / batchsize
                )
            )
            if (dx < TolX and abs(slope) < slope_err_max) or (abs(slope) < slope_err and dx < TolX_max):
                break
    x_out = np.mean(xtab[:, it - batchsize : it], axis=1)  # :96
    f_out = float(np.mean(ftab[it - batchsize : it]))
    return x_out.reshape(x0.shape), f_out, xtab[:, :it], ftab[:it], it


def _polyfit1_slope(x, y):
    """Slope of the degree-1 least-squares fit and A(1,1) of
    ``Rinv = inv(S.R); A = (Rinv*Rinv')*S.normr^2/S.df`` (utils/fminadam.m:67-69):
    the usual OLS variance of the slope, normr^2 / df / sum((x-mean x)^2)
    (x is centred here, so the closed form is exact)."""
    n = x.size
    xm = np.mean(x)
    sxx = np.sum((x - xm) ** 2)
    slope = np.sum((x - xm) * (y - np.mean(y))) / sxx
    icpt = np.mean(y) - slope * xm
    r = y - (slope * x + icpt)
    normr2 = float(np.sum(r * r))
    df = n - 2
    return float(slope), normr2 / df / sxx


# --------------------------------------------------------------------------
# sieve
# --------------------------------------------------------------------------


def gethpd_vbmc(X, y, HPDFrac=0.8):
    """misc/gethpd_vbmc.m:9-14 (stable descending sort like MATLAB's)."""
    N = X.shape[0]
    order = matlab_sort_descend(y)
    N_hpd = int(matlab_round(HPDFrac * N))
    return X[order[:N_hpd], :], y[order[:N_hpd]]


def matlab_round(x):
    return math.floor(abs(x) + 0.5) * (1 if x >= 0 else -1)


def matlab_sort_ascend(v):
    """MATLAB sort(v,'ascend'): stable, NaN placed last (in original order)."""
    v = np.asarray(v, dtype=np.float64)
    nan = np.isnan(v)
    good = np.nonzero(~nan)[0]
    return np.concatenate([good[np.argsort(v[good], kind="stable")], np.nonzero(nan)[0]])


def matlab_sort_descend(v):
    """MATLAB sort(v,'descend'): stable among ties (original order kept), NaN first."""
    v = np.asarray(v, dtype=np.float64)
    nan = np.isnan(v)
    idx = np.argsort(-np.where(nan, 0.0, v), kind="stable")
    idx = idx[~nan[idx]]
    return np.concatenate([np.nonzero(nan)[0], idx])


def sieve_order(nelcbo_fill):
    """misc/vpsieve_vbmc.m:81: ``[~,vp0_ord] = sort(nelcbo_fill,'ascend')``."""
    return matlab_sort_ascend(nelcbo_fill)


# --------------------------------------------------------------------------
# vbinit / vpsieve / vpoptimize (sequential, as the reference runs them)
# --------------------------------------------------------------------------

VBMC_OPTIONS = {  # the defaults of vbmc.m:158-366 that the path reads
    "TolLength": 1e-6, "TolWeight": 1e-2, "TolConLoss": 0.01, "WeightPenalty": 0.1, "HPDFrac": 0.8,
    "NSent": lambda K: 100 * K ** (2.0 / 3.0), "NSentFast": 0, "NSentFine": lambda K: 2**12 * K,
    "NSelbo": lambda K: 50 * K, "ELCBOWeight": 0, "SGDStepSize": 0.005, "TolFunStochastic": 1e-3,
    "MaxIterStochastic": None, "ELCBOmidpoint": True, "StochasticOptimizer": "adam", "TolImprovement": 0.01,
    "ELCBOImproWeight": 3, "PruningThresholdMultiplier": lambda K: 1.0 / math.sqrt(K), "VariationalInitRepo": False,
}


def evaloption_vbmc(option, N):
    """misc/evaloption_vbmc.m:4-8."""
    return option(N) if callable(option) else option


def _var_rows(A):
    """MATLAB var(A,[],2) of a D x n matrix (normalised by n-1) as a length-D vector."""
    return np.var(A, axis=1, ddof=1)


def vbinit_vbmc(type_, Nopts, vp, Knew, Xstar, ystar, rng):
    """misc/vbinit_vbmc.m:1-140 -> (list of vp, type vector).  MATLAB's randn / rand / randi / randperm become calls
    on ``rng`` (a numpy Generator) in the order the reference draws them: randn(1,Knew) -> standard_normal(Knew),
    randn(size(mu)) -> standard_normal((D,K)), randi(K) -> integers(K), rand() -> random(), randperm(n) -> permutation(n)."""
    D, K = vp["D"], vp["K"]
    Nstar = Xstar.shape[0]
    type_vec = type_ * np.ones(Nopts, dtype=np.int64)  # :17
    lambda0 = vp["lambda"].copy()
    mu0 = vp["mu"].copy()
    w0 = vp["w"].copy()
    if type_ == 1:  # :23-24
        sigma0 = vp["sigma"].copy()
    elif type_ == 2:  # :25-32
        ord_ = matlab_sort_descend(ystar)
        if vp["optimize_mu"]:
            idx_ord = np.tile(np.arange(min(Knew, Nstar)), int(math.ceil(Knew / Nstar)))
            mu0 = Xstar[ord_[idx_ord[:Knew]], :].T.copy()
        V = _var_rows(mu0) if K > 1 else np.var(Xstar, axis=0, ddof=1)
        sigma0 = np.sqrt(np.mean(V / lambda0**2) / Knew) * np.exp(0.2 * rng.standard_normal(Knew))
    elif type_ == 3:  # :33-35
        if vp["optimize_mu"]:
            mu0 = np.zeros((D, K))
        sigma0 = np.zeros(K)
    else:
        raise ValueError("vbinit:UnknownType")
    vp0_vec = []
    for iOpt in range(1, Nopts + 1):  # :38
        v = copy_vp(vp)
        v["K"] = Knew
        mu, sigma, lam = mu0.copy(), sigma0.copy(), lambda0.copy()
        w = w0.copy() if vp["optimize_weights"] else None
        add_jitter = True
        if type_ == 1:  # :50-73
            if iOpt == 1:
                add_jitter = False
            if Knew > vp["K"]:
                grow = Knew - vp["K"]
                mu = np.hstack([mu, np.zeros((D, grow))])
                sigma = np.concatenate([sigma, np.zeros(grow)])
                if w is not None:
                    w = np.concatenate([w, np.zeros(grow)])
                for iNew in range(vp["K"], Knew):
                    idx = int(rng.integers(vp["K"]))
                    mu[:, iNew] = mu[:, idx]
                    sigma[iNew] = sigma[idx]
                    mu[:, iNew] = mu[:, iNew] + 0.5 * sigma[iNew] * lam * rng.standard_normal(D)
                    if vp["optimize_sigma"]:
                        sigma[iNew] = sigma[iNew] * math.exp(0.2 * rng.standard_normal())
                    if vp["optimize_weights"]:
                        xi = 0.25 + 0.25 * rng.random()
                        w[iNew] = xi * w[idx]
                        w[idx] = (1 - xi) * w[idx]
        elif type_ == 2:  # :75-85
            if iOpt == 1:
                add_jitter = False
            if vp["optimize_lambda"]:
                lam = np.std(Xstar, axis=0, ddof=1)
                lam = lam * math.sqrt(D / np.sum(lam**2))
            if vp["optimize_weights"]:
                w = np.ones(Knew) / Knew
        else:  # :87-106
            ord_ = rng.permutation(Nstar)
            if vp["optimize_mu"]:
                idx_ord = np.tile(np.arange(min(Knew, Nstar)), int(math.ceil(Knew / Nstar)))
                mu = Xstar[ord_[idx_ord[:Knew]], :].T.copy()
            else:
                mu = mu0.copy()
            V = _var_rows(mu) if K > 1 else np.var(Xstar, axis=0, ddof=1)
            if vp["optimize_sigma"]:
                sigma = math.sqrt(np.mean(V) / Knew) * np.exp(0.2 * rng.standard_normal(Knew))
            if vp["optimize_lambda"]:
                lam = np.std(Xstar, axis=0, ddof=1)
                lam = lam * math.sqrt(D / np.sum(lam**2))
            if vp["optimize_weights"]:
                w = np.ones(Knew) / Knew
        if add_jitter:  # :113-127
            if vp["optimize_mu"]:
                mu = mu + sigma[None, :] * (lam[:, None] * rng.standard_normal(mu.shape))
            if vp["optimize_sigma"]:
                sigma = sigma * np.exp(0.2 * rng.standard_normal(Knew))
            if vp["optimize_lambda"]:
                lam = lam * np.exp(0.2 * rng.standard_normal(D))
            if vp["optimize_weights"]:
                w = w * np.exp(0.2 * rng.standard_normal(Knew))
                w = w / np.sum(w)
        v["w"] = w if vp["optimize_weights"] else np.ones(Knew) / Knew  # :129-133
        v["mu"] = mu if vp["optimize_mu"] else mu0.copy()
        v["sigma"] = sigma
        v["lambda"] = lam
        vp0_vec.append(v)
    return vp0_vec, type_vec


def vpsieve_vbmc(Ninit, Nbest, vp, gp, optimState, options, K=None, *, rng, eps_for=None):
    """misc/vpsieve_vbmc.m:1-90 -> (vp0_vec, vp0_type, elcbo_beta, compute_var, NSentK, NSentKFast, nelcbo_fill sorted input).
    ``eps_for(kind, slot, it, K, Ns)`` supplies the standard normals an MC-entropy evaluation consumes (kind 'sieve',
    slot = candidate index in generation order)."""
    optimState = dict(optimState or {})
    optimState.setdefault("delta", 0)
    optimState.setdefault("EntropySwitch", False)
    optimState.setdefault("Neff", gp["X"].shape[0])
    if Nbest is None:
        Nbest = 1
    if K is None:
        K = vp["K"]
    vp = copy_vp(vp)
    vp["delta"] = optimState["delta"]  # :18
    if Ninit is None:
        Ninit = int(math.ceil(evaloption_vbmc(options["NSelbo"], K)))
    NSentK = int(math.ceil(evaloption_vbmc(options["NSent"], K) / K))  # :26
    NSentKFast = int(math.ceil(evaloption_vbmc(options["NSentFast"], K) / K))
    if optimState["EntropySwitch"] or K == 1:  # :30-33
        NSentK = NSentKFast = 0
    elcbo_beta = evaloption_vbmc(options["ELCBOWeight"], optimState["Neff"])  # :36
    compute_var = elcbo_beta != 0
    vp, thetabnd = vpbounds(vp, gp, options, K)  # :40
    if Ninit > 0:
        Xstar, ystar = gethpd_vbmc(gp["X"], gp["y"], options["HPDFrac"])  # :46
        if Nbest == 1:
            vp0_vec, vp0_type = vbinit_vbmc(1, Ninit, vp, K, Xstar, ystar, rng)
        else:
            n3 = int(math.ceil(Ninit / 3))
            v1, t1 = vbinit_vbmc(1, n3, vp, K, Xstar, ystar, rng)
            v2, t2 = vbinit_vbmc(2, n3, vp, K, Xstar, ystar, rng)
            v3, t3 = vbinit_vbmc(3, Ninit - 2 * n3, vp, K, Xstar, ystar, rng)
            vp0_vec = v1 + v2 + v3
            vp0_type = np.concatenate([t1, t2, t3])
        repo = optimState.get("vp_repo")
        if repo and options.get("VariationalInitRepo"):  # :62-72
            Ntheta = get_vptheta(vp0_vec[0])[0].size
            extra = [rescale_params(vp0_vec[0], th) for th in repo if np.size(th) == Ntheta]
            vp0_vec = vp0_vec + extra
            vp0_type = np.concatenate([vp0_type, np.ones(len(extra), dtype=np.int64)])
        nelcbo_fill = np.zeros(len(vp0_vec))
        for iOpt in range(len(vp0_vec)):  # :75-79
            theta0, vp0_vec[iOpt] = get_vptheta(vp0_vec[iOpt], vp["optimize_mu"], vp["optimize_sigma"], vp["optimize_lambda"],
                                                vp["optimize_weights"])
            eps = eps_for("sieve", iOpt, 0, K, NSentKFast) if (NSentKFast > 0 and eps_for) else None
            r = negelcbo_vbmc(theta0, 0, vp0_vec[iOpt], gp, NSentKFast, False, int(compute_var), thetabnd=thetabnd, eps=eps,
                              rng=None if eps is not None else rng)
            nelcbo_fill[iOpt] = r["F"] + elcbo_beta * math.sqrt(r["varF"])
        order = matlab_sort_ascend(nelcbo_fill)  # :82
        vp0_vec = [vp0_vec[i] for i in order]
        vp0_type = vp0_type[order]
    else:
        vp0_vec, vp0_type, nelcbo_fill = [vp], np.array([1]), np.zeros(0)
    return vp0_vec, vp0_type, elcbo_beta, compute_var, NSentK, NSentKFast, nelcbo_fill


def eval_fullelcbo(theta, vp, gp, beta, options, eps=None, rng=None):
    """eval_fullelcbo, misc/vpoptimize_vbmc.m:257-305 (the filling branch) -> dict of the fields it stores."""
    K = vp["K"]
    NSentFineK = int(math.ceil(evaloption_vbmc(options["NSentFine"], K) / K))  # :278
    computevar_flag = not options.get("SkipELBOVariance", False)
    theta = np.asarray(theta, dtype=np.float64).reshape(-1)
    r = negelcbo_vbmc(theta, 0, vp, gp, NSentFineK, False, 1 if computevar_flag else 0, thetabnd=None, separate_K=True, eps=eps,
                      rng=rng)  # :288-289
    return {"nelbo": r["F"], "G": r["G"], "H": r["H"], "varF": r["varF"], "varG": r["varG"], "varH": r["varH"],
            "varss": r["varGss"], "nelcbo": r["F"] + beta * math.sqrt(r["varF"]), "theta": theta.copy(), "I_sk": r["I_sk"],
            "J_sjk": r["J_sjk"]}


def vpoptimize_vbmc(Nfastopts, Nslowopts, vp, gp, K=None, optimState=None, options=None, *, rng, eps_for):
    """misc/vpoptimize_vbmc.m:1-254, the gradient-available stochastic branch (ELCBOWeight = 0, NSentK > 0, Adam :108-135)
    plus the pruning loop (:196-243) -> (vp, varss, pruned).  ``eps_for(kind, slot, it, K, Ns)`` supplies the draws of
    every MC-entropy evaluation: kind 'adam' (slot = chain iOpt-1, it = Adam iteration 1..), 'full' (slot = 2*(iOpt-1) for
    the midpoint, +1 for the endpoint), 'prune' (slot = running count 1..)."""
    options = dict(VBMC_OPTIONS, **(options or {}))
    optimState = dict(optimState or {})
    if K is None:
        K = vp["K"]
    optimState.setdefault("Warmup", not vp["optimize_weights"])  # :18
    optimState.setdefault("temperature", 1)
    vp0_vec, vp0_type, elcbo_beta, compute_var, NSentK, _, _ = vpsieve_vbmc(Nfastopts, Nslowopts, vp, gp, optimState, options, K,
                                                                            rng=rng, eps_for=eps_for)  # :23-24
    vp, thetabnd = vpbounds(vp, gp, options, K)  # :27
    if compute_var or NSentK == 0:
        raise NotImplementedError("only the Adam branch is restated (CMA-ES / fminunc are third-party optimisers)")
    D = vp["D"]
    vp0_type = list(vp0_type)
    nslot = 2 * Nslowopts
    stats = [None] * nslot
    nelcbo = np.full(nslot, np.inf)  # :262-269
    vp0_fine = [None] * nslot
    for iOpt in range(1, Nslowopts + 1):  # :49
        iOpt_mid, iOpt_end = 2 * iOpt - 2, 2 * iOpt - 1
        if Nslowopts == 1:  # :54-61
            idx = 0
        elif Nslowopts == 2:
            idx = [i for i, t in enumerate(vp0_type) if (t == 1 if iOpt == 1 else (t == 2 or t == 3))][0]
        else:
            idx = [i for i, t in enumerate(vp0_type) if t == ((iOpt - 1) % 3) + 1][0]
        vp0 = rescale_params(vp0_vec[idx])  # :65
        vp0_type.pop(idx)
        vp0_vec.pop(idx)
        parts = []  # :68-71
        if vp["optimize_mu"]:
            parts.append(vp0["mu"].reshape(-1, order="F"))
        if vp["optimize_sigma"]:
            parts.append(np.log(vp0["sigma"]))
        if vp["optimize_lambda"]:
            parts.append(np.log(vp0["lambda"]))
        if vp["optimize_weights"]:
            parts.append(np.log(vp0["w"]))
        theta0 = np.concatenate(parts)
        it_box = [0]

        def vbtrainmc_fun(theta_, vp0=vp0, iOpt=iOpt):  # :74
            it_box[0] += 1
            r = negelcbo_vbmc(theta_, elcbo_beta, vp0, gp, NSentK, True, 0, thetabnd=thetabnd,
                              eps=eps_for("adam", iOpt - 1, it_box[0], K, NSentK))
            return r["F"], r["dF"]

        ms = {"min": min(options["SGDStepSize"], 0.001)}  # :112-127
        if optimState["Warmup"] or not vp["optimize_weights"]:
            scaling_factor = min(0.1, options["SGDStepSize"] * 10)
        else:
            scaling_factor = min(0.1, options["SGDStepSize"])
        ms["max"] = max(ms["min"], scaling_factor)
        ms["decay"] = 200
        MaxIter = int(min(options["MaxIterStochastic"] or 100 * (2 + D), 1e4))
        thetaopt, _, theta_lst, fval_lst, _ = fminadam(vbtrainmc_fun, theta0, TolFun=options["TolFunStochastic"], MaxIter=MaxIter,
                                                       master_stepsize=ms)  # :128-129
        if options["ELCBOmidpoint"]:  # :131-136
            idx_mid = int(np.argmin(fval_lst))
            stats[iOpt_mid] = eval_fullelcbo(theta_lst[:, idx_mid], vp0, gp, elcbo_beta, options,
                                             eps=eps_for("full", iOpt_mid, 0, K, None))
            nelcbo[iOpt_mid] = stats[iOpt_mid]["nelcbo"]
        stats[iOpt_end] = eval_fullelcbo(thetaopt, vp0, gp, elcbo_beta, options, eps=eps_for("full", iOpt_end, 0, K, None))  # :165
        nelcbo[iOpt_end] = stats[iOpt_end]["nelcbo"]
        vp0_fine[iOpt_mid] = vp0
        vp0_fine[iOpt_end] = vp0
    idx = int(np.argmin(nelcbo))  # :176
    s = stats[idx]
    elbo = -s["nelbo"]
    elbo_sd = math.sqrt(s["varF"])
    G, H, varss, varG, varH = s["G"], s["H"], s["varss"], s["varG"], s["varH"]
    I_sk = np.array(s["I_sk"], copy=True)
    J_sjk = np.array(s["J_sjk"], copy=True)
    vp = rescale_params(vp0_fine[idx], s["theta"])  # :189-190
    vp["temperature"] = optimState["temperature"]
    pruned = 0  # :196
    if vp["optimize_weights"]:
        alreadychecked = np.zeros(vp["K"], dtype=bool)
        count = 0
        while np.any((vp["w"] < options["TolWeight"]) & ~alreadychecked):  # :201
            vp_pruned = copy_vp(vp)
            cand = np.nonzero((vp_pruned["w"] < options["TolWeight"]) & ~alreadychecked)[0]
            idx = int(cand[int(rng.integers(cand.size))])  # :206
            vp_pruned["w"] = np.delete(vp_pruned["w"], idx)
            if "eta" in vp_pruned:
                vp_pruned["eta"] = np.delete(vp_pruned["eta"], idx)
            vp_pruned["sigma"] = np.delete(vp_pruned["sigma"], idx)
            vp_pruned["mu"] = np.delete(vp_pruned["mu"], idx, axis=1)
            vp_pruned["K"] -= 1
            theta_pruned, vp_pruned = get_vptheta(vp_pruned, vp_pruned["optimize_mu"], vp_pruned["optimize_sigma"],
                                                  vp_pruned["optimize_lambda"], vp_pruned["optimize_weights"])  # :212
            count += 1
            sp = eval_fullelcbo(theta_pruned, vp_pruned, gp, elcbo_beta, options, eps=eps_for("prune", count, 0, vp_pruned["K"], None))
            elbo_pruned = -sp["nelbo"]
            elbo_pruned_sd = math.sqrt(sp["varF"])
            delta_elcbo = abs((elbo_pruned - options["ELCBOImproWeight"] * elbo_pruned_sd)
                              - (elbo - options["ELCBOImproWeight"] * elbo_sd))  # :220-221
            PruningThreshold = options["TolImprovement"] * evaloption_vbmc(options["PruningThresholdMultiplier"], K)  # :224-225
            if delta_elcbo < PruningThreshold:
                vp = vp_pruned
                elbo, elbo_sd = elbo_pruned, elbo_pruned_sd
                G, H, varss, varG, varH = sp["G"], sp["H"], sp["varss"], sp["varG"], sp["varH"]
                pruned += 1
                alreadychecked = np.delete(alreadychecked, idx)
                I_sk = np.delete(I_sk, idx, axis=1)  # :238
                J_sjk = np.delete(J_sjk, idx, axis=2)  # :239: third dimension only
            else:
                alreadychecked[idx] = True
    vp["stats"] = {"elbo": elbo, "elbo_sd": elbo_sd, "elogjoint": G, "elogjoint_sd": math.sqrt(varG), "entropy": H,
                   "entropy_sd": math.sqrt(varH), "stable": False, "I_sk": I_sk, "J_sjk": J_sjk}
    return vp, varss, pruned
