"""Generate tests/golden/mp_*.json: 50-digit mpmath evaluation of the hot-path formulas.

TEST INFRASTRUCTURE.  This is an *independent* scalar-loop restatement (no NumPy
broadcasting, no code shared with oracle/vbmc_ref.py) of the reference formulas,
evaluated with mpmath at 50 significant digits on tiny shapes.  Its outputs are the
committed golden vectors that pin the NumPy/C oracles and the HIP path to ~1e-12.

Formulas follow (reference paths, acerbilab/vbmc v1.0.12):
  entmc    ent/entmc_vbmc.m:49-125
  entlb    ent/entlb_vbmc.m:66-141
  logjoint misc/gplogjoint.m:97-413 (negquad mean, meanfun id 4; also 0 and 1)
  gp_post  gplite/private/gplite_core.m:33-102,278-291
  gp_pred  gplite/gplite_pred.m:52-165
  pred     gplite/gplite_noisefun.m:176-210 + gplite_core.m:33-102 + gplite_pred.m:60-127 with the general noise models,
           ystar / s2star and the log predictive density lp (mp_pred_case*.json)
  pen      misc/vpbndloss.m:1-73, utils/softbndloss.m:1-30, misc/negelcbo_vbmc.m:146-162 (mp_pen_case*.json)
  nlZ      gplite/private/gplite_core.m:205 (value); its gradient (:236-275) is pinned by 50-digit central
           differences of the value, i.e. independently of the reference's analytic Q-matrix formulas

Run:  python oracle/mp_golden.py   (writes tests/golden/mp_case*.json)
The inputs are drawn with numpy default_rng(seed) and stored in the JSON next to
the expected outputs, so the fixtures are self-contained data.
"""
from __future__ import annotations

import json
import math
import os
import sys

import mpmath as mp
import numpy as np

mp.mp.dps = 50


def M(x):
    return mp.mpf(float(x))


def softmax_jac(eta):
    K = len(eta)
    e = [mp.e ** t for t in eta]
    ssum = mp.fsum(e)
    J = [[-(e[a] * e[b]) / ssum**2 + (e[a] / ssum if a == b else 0) for b in range(K)] for a in range(K)]
    return J


def matvec(J, v):
    return [mp.fsum(J[a][b] * v[b] for b in range(len(v))) for a in range(len(J))]


# ------------------------------------------------------------------ entmc
def mp_entmc(mu, sigma, lam, w, eta, eps):
    """mu[d][k], sigma[k], lam[d], w[k], eta[k], eps[j][i][d] (i < M/2)."""
    D, K = len(mu), len(sigma)
    Mh = len(eps[0])
    Ns = 2 * Mh
    nf = 1 / (2 * mp.pi) ** (mp.mpf(D) / 2) / mp.fprod(lam)
    H = mp.mpf(0)
    mu_g = [[mp.mpf(0)] * K for _ in range(D)]
    sg_g = [mp.mpf(0)] * K
    lam_g = [mp.mpf(0)] * D
    w_g = [mp.mpf(0)] * K
    for j in range(K):
        samples = [eps[j][i] for i in range(Mh)] + [[-e for e in eps[j][i]] for i in range(Mh)]
        for e in samples:
            x = [e[d] * lam[d] * sigma[j] + mu[d][j] for d in range(D)]
            norm = []
            for k in range(K):
                d2 = mp.fsum(((x[d] - mu[d][k]) / (sigma[k] * lam[d])) ** 2 for d in range(D))
                norm.append(nf / sigma[k] ** D * mp.e ** (-d2 / 2))
            q = mp.fsum(w[k] * norm[k] for k in range(K))
            H -= w[j] * mp.log(q) / Ns
            lsum = [mp.fsum((x[d] - mu[d][k]) / (sigma[k] * lam[d]) ** 2 * norm[k] * w[k] for k in range(K)) for d in range(D)]
            for d in range(D):
                mu_g[d][j] += w[j] * lsum[d] / q / Ns
                lam_g[d] += lsum[d] * w[j] * sigma[j] * e[d] / q / Ns
            sg_g[j] += w[j] * mp.fsum(lsum[d] * e[d] * lam[d] for d in range(D)) / q / Ns
            w_g[j] -= mp.log(q) / Ns
            for l in range(K):
                w_g[l] -= w[j] * norm[l] / q / Ns
    lam_g = [lam_g[d] * lam[d] for d in range(D)]
    sg_g = [sg_g[k] * sigma[k] for k in range(K)]
    w_g = matvec(softmax_jac(eta), w_g)
    dH = [mu_g[d][k] for k in range(K) for d in range(D)] + sg_g + lam_g + w_g
    return H, dH


# ------------------------------------------------------------------ entlb
def mp_entlb(mu, sigma, lam, w, eta):
    D, K = len(mu), len(sigma)
    nconst = 1 / (2 * mp.pi) ** (mp.mpf(D) / 2) / mp.fprod(lam)

    def Hfun(mu, sigma, lam, w):
        nconst = 1 / (2 * mp.pi) ** (mp.mpf(D) / 2) / mp.fprod(lam)
        H = mp.mpf(0)
        for n in range(K):
            gs = mp.mpf(0)
            for k in range(K):
                ss2 = sigma[n] ** 2 + sigma[k] ** 2
                d2 = mp.fsum((mu[d][n] - mu[d][k]) ** 2 / (ss2 * lam[d] ** 2) for d in range(D))
                gs += w[k] * nconst * ss2 ** (-mp.mpf(D) / 2) * mp.e ** (-d2 / 2)
            H -= w[n] * mp.log(gs)
        return H

    H = Hfun(mu, sigma, lam, w)

    # Gradient by high-precision differentiation of H in the *transformed* parameters
    # (valid: the entlb gradient in the reference is the exact derivative of the bound).
    def Htheta(th):
        mu_ = [[th[k * D + d] for k in range(K)] for d in range(D)]
        sg_ = [mp.e ** th[D * K + k] for k in range(K)]
        lm_ = [mp.e ** th[D * K + K + d] for d in range(D)]
        et_ = th[D * K + K + D :]
        ee = [mp.e ** t for t in et_]
        sm = mp.fsum(ee)
        w_ = [t / sm for t in ee]
        return Hfun(mu_, sg_, lm_, w_)

    th0 = [mu[d][k] for k in range(K) for d in range(D)] + [mp.log(s) for s in sigma] + [mp.log(l) for l in lam] + list(eta)
    dH = []
    for i in range(len(th0)):
        dH.append(mp.diff(lambda t: Htheta(th0[:i] + [t] + th0[i + 1 :]), th0[i]))
    return H, dH


# ------------------------------------------------------------------ GP pieces
def mp_chol_upper(A):
    n = len(A)
    R = [[mp.mpf(0)] * n for _ in range(n)]
    for j in range(n):
        s = A[j][j] - mp.fsum(R[i][j] ** 2 for i in range(j))
        R[j][j] = mp.sqrt(s)
        for c in range(j + 1, n):
            R[j][c] = (A[j][c] - mp.fsum(R[i][j] * R[i][c] for i in range(j))) / R[j][j]
    return R


def mp_solve_ut_t(R, b):  # R' \ b
    n = len(b)
    x = [mp.mpf(0)] * n
    for i in range(n):
        x[i] = (b[i] - mp.fsum(R[r][i] * x[r] for r in range(i))) / R[i][i]
    return x


def mp_solve_ut(R, b):  # R \ b
    n = len(b)
    x = [mp.mpf(0)] * n
    for i in reversed(range(n)):
        x[i] = (b[i] - mp.fsum(R[i][c] * x[c] for c in range(i + 1, n))) / R[i][i]
    return x


def mp_meanfun(hyp_mean, x, meanfun, D):
    if meanfun == 0:
        return mp.mpf(0)
    if meanfun == 1:
        return hyp_mean[0]
    m0 = hyp_mean[0]
    xm = hyp_mean[1 : D + 1]
    om = [mp.e ** t for t in hyp_mean[D + 1 : 2 * D + 1]]
    return m0 - mp.fsum(((x[d] - xm[d]) / om[d]) ** 2 for d in range(D)) / 2


def mp_gp_post(hyp, X, y, meanfun):
    """One hyper-sample; const noise (noisefun [1 0 0]); returns alpha, L (upper), sn2."""
    N, D = len(X), len(X[0])
    ell = [mp.e ** hyp[d] for d in range(D)]
    sf2 = mp.e ** (2 * hyp[D])
    sn2 = mp.e ** (2 * hyp[D + 1])
    hyp_mean = hyp[D + 2 :]
    Kmat = [[sf2 * mp.e ** (-mp.fsum(((X[a][d] - X[b][d]) / ell[d]) ** 2 for d in range(D)) / 2) for b in range(N)] for a in range(N)]
    A = [[Kmat[a][b] / sn2 + (1 if a == b else 0) for b in range(N)] for a in range(N)]
    L = mp_chol_upper(A)
    r = [y[n] - mp_meanfun(hyp_mean, X[n], meanfun, D) for n in range(N)]
    alpha = [t / sn2 for t in mp_solve_ut(L, mp_solve_ut_t(L, r))]
    return alpha, L, sn2


def mp_gp_pred(hyp, X, alpha, L, sn2, Xs, meanfun):
    N, D = len(X), len(X[0])
    ell = [mp.e ** hyp[d] for d in range(D)]
    sf2 = mp.e ** (2 * hyp[D])
    hyp_mean = hyp[D + 2 :]
    fmu, fs2 = [], []
    for xs in Xs:
        ks = [sf2 * mp.e ** (-mp.fsum(((X[n][d] - xs[d]) / ell[d]) ** 2 for d in range(D)) / 2) for n in range(N)]
        fmu.append(mp_meanfun(hyp_mean, xs, meanfun, D) + mp.fsum(ks[n] * alpha[n] for n in range(N)))
        v = mp_solve_ut_t(L, [k / mp.sqrt(sn2) for k in ks])
        fs2.append(max(sf2 - mp.fsum(t * t for t in v), mp.mpf(0)))
    return fmu, fs2


# ------------------------------------------------------------------ gplogjoint
def mp_logjoint(mu, sigma, lam, w, eta, X, posts, meanfun, compute_var):
    """posts: list of dict(hyp, alpha, L, sn2).  Returns per-sample F[s], dF[s][t],
    I_sk, J_sjk (full), varF[s] for compute_var in {1,2}; averaging done by caller tests."""
    D, K, N = len(mu), len(sigma), len(X)
    S = len(posts)
    eps = mp.mpf(2) ** -52
    Fs, dFs, I_sk, J_all, varFs = [], [], [], [], []
    Jw = softmax_jac(eta)
    for s in range(S):
        hyp = posts[s]["hyp"]
        alpha = posts[s]["alpha"]
        L = posts[s]["L"]
        sn2 = posts[s]["sn2"]
        ell = [mp.e ** hyp[d] for d in range(D)]
        ln_sf2 = 2 * hyp[D]
        sum_lnell = mp.fsum(hyp[:D])
        hm = hyp[D + 2 :]
        m0 = hm[0] if meanfun > 0 else mp.mpf(0)
        if meanfun == 4:
            xm = hm[1 : D + 1]
            om = [mp.e ** t for t in hm[D + 1 : 2 * D + 1]]
        F = mp.mpf(0)
        mu_g = [[mp.mpf(0)] * K for _ in range(D)]
        sg_g = [mp.mpf(0)] * K
        lam_g = [mp.mpf(0)] * D
        w_g = [mp.mpf(0)] * K
        Ik = []
        zs = []
        for k in range(K):
            tau = [mp.sqrt(sigma[k] ** 2 * lam[d] ** 2 + ell[d] ** 2) for d in range(D)]
            lnnf = ln_sf2 + sum_lnell - mp.fsum(mp.log(t) for t in tau)
            delt = [[(mu[d][k] - X[n][d]) / tau[d] for n in range(N)] for d in range(D)]
            z = [mp.e ** (lnnf - mp.fsum(delt[d][n] ** 2 for d in range(D)) / 2) for n in range(N)]
            zs.append(z)
            I = mp.fsum(z[n] * alpha[n] for n in range(N)) + m0
            if meanfun == 4:
                I += -mp.fsum((mu[d][k] ** 2 + sigma[k] ** 2 * lam[d] ** 2 - 2 * mu[d][k] * xm[d] + xm[d] ** 2) / om[d] ** 2 for d in range(D)) / 2
            Ik.append(I)
            F += w[k] * I
            for d in range(D):
                mu_g[d][k] = w[k] * mp.fsum(-delt[d][n] / tau[d] * z[n] * alpha[n] for n in range(N))
                if meanfun == 4:
                    mu_g[d][k] -= w[k] / om[d] ** 2 * (mu[d][k] - xm[d])
            sg_g[k] = w[k] * mp.fsum(
                mp.fsum((lam[d] / tau[d]) ** 2 * (delt[d][n] ** 2 - 1) for d in range(D)) * sigma[k] * z[n] * alpha[n] for n in range(N)
            )
            if meanfun == 4:
                sg_g[k] -= w[k] * sigma[k] * mp.fsum(lam[d] ** 2 / om[d] ** 2 for d in range(D))
            for d in range(D):
                lam_g[d] += w[k] * mp.fsum((sigma[k] / tau[d]) ** 2 * (delt[d][n] ** 2 - 1) * lam[d] * z[n] * alpha[n] for n in range(N))
                if meanfun == 4:
                    lam_g[d] -= w[k] * sigma[k] ** 2 / om[d] ** 2 * lam[d]
            w_g[k] = I
        sg_g = [sg_g[k] * sigma[k] for k in range(K)]
        lam_g = [lam_g[d] * lam[d] for d in range(D)]
        w_g = matvec(Jw, w_g)
        Fs.append(F)
        dFs.append([mu_g[d][k] for k in range(K) for d in range(D)] + sg_g + lam_g + w_g)
        I_sk.append(Ik)
        if compute_var:
            # K^{-1} z_j  (Lchol branch: (L\(L'\z))/sn2_eff)
            Kinvz = [[t / sn2 for t in mp_solve_ut(L, mp_solve_ut_t(L, zs[j]))] for j in range(K)]
            J = [[mp.mpf(0)] * K for _ in range(K)]
            for k in range(K):
                for j in range(K):
                    tau_jk = [mp.sqrt((sigma[j] ** 2 + sigma[k] ** 2) * lam[d] ** 2 + ell[d] ** 2) for d in range(D)]
                    lnnf_jk = ln_sf2 + sum_lnell - mp.fsum(mp.log(t) for t in tau_jk)
                    d_jk = mp.fsum(((mu[d][j] - mu[d][k]) / tau_jk[d]) ** 2 for d in range(D))
                    J[j][k] = mp.e ** (lnnf_jk - d_jk / 2) - mp.fsum(zs[k][n] * Kinvz[j][n] for n in range(N))
            J_all.append(J)
            if compute_var == 2:
                v = mp.fsum(w[k] ** 2 * max(eps, J[k][k]) for k in range(K))
            else:
                v = mp.fsum(w[k] ** 2 * max(eps, J[k][k]) for k in range(K)) + mp.fsum(
                    2 * w[j] * w[k] * J[j][k] for k in range(K) for j in range(k)
                )
            varFs.append(max(v, eps))
    return Fs, dFs, I_sk, J_all, varFs


# ------------------------------------------------------------------ GP marginal likelihood
def mp_nlz(hyp, X, y, meanfun, noisefun, s2):
    """-log N(y; m, K + diag(sn2)) straight from the definition (no sn2 rescaling, no retries)."""
    N, D = len(X), len(X[0])
    ell = [mp.e ** hyp[d] for d in range(D)]
    sf2 = mp.e ** (2 * hyp[D])
    idx = D + 1
    sn2 = [mp.e ** (2 * hyp[idx])] * N
    idx += 1
    if noisefun[1] == 1:
        sn2 = [sn2[n] + s2[n] for n in range(N)]
    elif noisefun[1] == 2:
        sn2 = [sn2[n] + mp.e ** hyp[idx] * s2[n] for n in range(N)]
        idx += 1
    hyp_mean = hyp[idx:]
    A = [[sf2 * mp.e ** (-mp.fsum(((X[a][d] - X[b][d]) / ell[d]) ** 2 for d in range(D)) / 2) + (sn2[a] if a == b else 0)
          for b in range(N)] for a in range(N)]
    R = mp_chol_upper(A)
    r = [y[n] - mp_meanfun(hyp_mean, X[n], meanfun, D) for n in range(N)]
    v = mp_solve_ut_t(R, r)
    return mp.fsum(t * t for t in v) / 2 + mp.fsum(mp.log(R[i][i]) for i in range(N)) + N * mp.log(2 * mp.pi) / 2


def mp_nlz_grad(hyp, X, y, meanfun, noisefun, s2):
    h = mp.mpf(10) ** -18
    g = []
    for i in range(len(hyp)):
        hp = list(hyp)
        hm = list(hyp)
        hp[i] = hyp[i] + h
        hm[i] = hyp[i] - h
        g.append((mp_nlz(hp, X, y, meanfun, noisefun, s2) - mp_nlz(hm, X, y, meanfun, noisefun, s2)) / (2 * h))
    return g


NLZ_CASES = [
    dict(seed=21, D=2, N=7, S=2, meanfun=4, noisefun=(1, 0, 0)),
    dict(seed=22, D=3, N=10, S=2, meanfun=4, noisefun=(1, 1, 0)),
    dict(seed=23, D=2, N=8, S=2, meanfun=1, noisefun=(1, 2, 0)),
    dict(seed=24, D=3, N=9, S=1, meanfun=0, noisefun=(1, 0, 0)),
]


def make_nlz_case(seed, D, N, S, meanfun, noisefun):
    rng = np.random.default_rng(seed)
    X = 1.5 * rng.standard_normal((N, D))
    y = -0.5 * np.sum((X / 1.3) ** 2, axis=1) + 0.3 * np.sin(X[:, 0]) + 0.05 * rng.standard_normal(N)
    s2 = 0.01 + 0.05 * rng.random(N) if noisefun[1] else None
    nnoise = 1 + (1 if noisefun[1] == 2 else 0)
    nmean = {0: 0, 1: 1, 4: 2 * D + 1}[meanfun]
    hyp = np.zeros((D + 1 + nnoise + nmean, S))
    for s in range(S):
        hyp[:D, s] = np.log(0.8) + 0.2 * rng.standard_normal(D)
        hyp[D, s] = np.log(np.std(y)) + 0.1 * rng.standard_normal()
        hyp[D + 1, s] = np.log(5e-2) + 0.1 * rng.standard_normal()
        if noisefun[1] == 2:
            hyp[D + 2, s] = 0.3 * rng.standard_normal()
        o = D + 1 + nnoise
        if meanfun >= 1:
            hyp[o, s] = np.max(y) + 0.1 * rng.standard_normal()
        if meanfun == 4:
            hyp[o + 1 : o + 1 + D, s] = 0.2 * rng.standard_normal(D)
            hyp[o + 1 + D :, s] = np.log(2.0) + 0.1 * rng.standard_normal(D)
    return dict(seed=seed, D=D, N=N, S=S, meanfun=meanfun, noisefun=list(noisefun), X=X, y=y, s2=s2, hyp=hyp)


def run_nlz_case(c):
    N, D = c["X"].shape
    X = [[M(c["X"][n, d]) for d in range(D)] for n in range(N)]
    y = [M(t) for t in c["y"]]
    s2 = None if c["s2"] is None else [M(t) for t in c["s2"]]
    out = {"nlZ": [], "dnlZ": []}
    for s in range(c["S"]):
        hyp = [M(t) for t in c["hyp"][:, s]]
        out["nlZ"].append(fl(mp_nlz(hyp, X, y, c["meanfun"], c["noisefun"], s2)))
        out["dnlZ"].append(fl(mp_nlz_grad(hyp, X, y, c["meanfun"], c["noisefun"], s2)))
    return out



# ------------------------------------------------------------------ prediction with the general noise models
def mp_noise(hyp_noise, noisefun, yv, s2v, n):
    """gplite/gplite_noisefun.m:176-210 for n points: constant + provided (scaled) s2 + output-dependent term."""
    idx = 0
    if noisefun[0] == 1:
        sn2 = [mp.e ** (2 * hyp_noise[idx])] * n
        idx += 1
    else:
        sn2 = [mp.mpf(2) ** -52] * n
    if noisefun[1] == 1:
        sn2 = [sn2[i] + s2v[i] for i in range(n)]
    elif noisefun[1] == 2:
        sn2 = [sn2[i] + mp.e ** hyp_noise[idx] * s2v[i] for i in range(n)]
        idx += 1
    if noisefun[2] == 1 and yv is not None:
        yth, w2 = hyp_noise[idx], mp.e ** (2 * hyp_noise[idx + 1])
        sn2 = [sn2[i] + w2 * max(mp.mpf(0), yth - yv[i]) ** 2 for i in range(n)]
    return sn2


def mp_pred_general(hyp, X, y, s2, Xs, ystar, s2star, meanfun, noisefun):
    """gplite_post + gplite_pred from the definitions (gplite_core.m:33-102 is algebraically alpha = (K + diag(sn2))^-1 (y - m)
    in BOTH its Lchol branches, gplite_pred.m:83-121 is fmu = m* + ks' alpha, fs2 = kss - ks' (K + diag(sn2))^-1 ks,
    ys2 = fs2 + sn2*, lp the Gaussian log density of ystar): one hyper-sample, 50 digits, no jitter retries (sn2_mult = 1)."""
    N, D = len(X), len(X[0])
    nn = int(noisefun[0] == 1) + int(noisefun[1] == 2) + 2 * int(noisefun[2] == 1)
    ell = [mp.e ** hyp[d] for d in range(D)]
    sf2 = mp.e ** (2 * hyp[D])
    hyp_noise, hyp_mean = hyp[D + 1 : D + 1 + nn], hyp[D + 1 + nn :]
    sn2 = mp_noise(hyp_noise, noisefun, y, s2, N)
    kern = lambda p, q: sf2 * mp.e ** (-mp.fsum(((p[d] - q[d]) / ell[d]) ** 2 for d in range(D)) / 2)  # noqa: E731
    A = [[kern(X[a], X[b]) + (sn2[a] if a == b else 0) for b in range(N)] for a in range(N)]
    R = mp_chol_upper(A)
    r = [y[n] - mp_meanfun(hyp_mean, X[n], meanfun, D) for n in range(N)]
    alpha = mp_solve_ut(R, mp_solve_ut_t(R, r))
    sn2s = mp_noise(hyp_noise, noisefun, ystar, s2star, len(Xs))
    fmu, fs2, ys2, lp = [], [], [], []
    for i, xs in enumerate(Xs):
        ks = [kern(X[n], xs) for n in range(N)]
        fmu.append(mp_meanfun(hyp_mean, xs, meanfun, D) + mp.fsum(ks[n] * alpha[n] for n in range(N)))
        v = mp_solve_ut_t(R, ks)
        fs2.append(max(sf2 - mp.fsum(t * t for t in v), mp.mpf(0)))
        ys2.append(fs2[-1] + sn2s[i])
        lp.append(-(ystar[i] - fmu[-1]) ** 2 / ys2[-1] / 2 - mp.log(2 * mp.pi * ys2[-1]) / 2)
    return alpha, fmu, fs2, ys2, lp, min(sn2)


PRED_CASES = [
    # tiny constant noise: min(sn2) < 1e-6 -> the Lchol = false branch (L = -inv(K + sn2 I), gplite_core.m:84-98)
    dict(seed=31, D=2, N=8, S=2, meanfun=4, noisefun=(1, 0, 0), logsn=math.log(5e-4)),
    dict(seed=32, D=3, N=9, S=2, meanfun=1, noisefun=(1, 1, 0), logsn=math.log(5e-2)),   # provided s2 (heteroscedastic Lchol branch)
    dict(seed=33, D=2, N=7, S=1, meanfun=4, noisefun=(1, 2, 0), logsn=math.log(3e-2)),   # scaled s2
    dict(seed=34, D=3, N=10, S=2, meanfun=0, noisefun=(1, 1, 1), logsn=math.log(5e-2)),  # + output-dependent noise (needs ystar)
    dict(seed=35, D=2, N=9, S=2, meanfun=4, noisefun=(1, 0, 1), logsn=math.log(2e-2)),
]


def make_pred_case(seed, D, N, S, meanfun, noisefun, logsn):
    rng = np.random.default_rng(seed)
    X = 1.5 * rng.standard_normal((N, D))
    y = -0.5 * np.sum((X / 1.3) ** 2, axis=1) + 0.3 * np.sin(X[:, 0]) + 0.05 * rng.standard_normal(N)
    s2 = 0.01 + 0.05 * rng.random(N) if noisefun[1] else None
    Ns = 5
    Xstar = 1.2 * rng.standard_normal((Ns, D))
    Xstar[0] = X[1] + 1e-3                       # next to a training input
    ystar = -0.5 * np.sum((Xstar / 1.3) ** 2, axis=1) + 0.2 * rng.standard_normal(Ns)
    s2star = 0.01 + 0.05 * rng.random(Ns) if noisefun[1] else None
    nnoise = int(noisefun[0] == 1) + int(noisefun[1] == 2) + 2 * int(noisefun[2] == 1)
    nmean = {0: 0, 1: 1, 4: 2 * D + 1}[meanfun]
    hyp = np.zeros((D + 1 + nnoise + nmean, S))
    for s in range(S):
        hyp[:D, s] = np.log(0.8) + 0.2 * rng.standard_normal(D)
        hyp[D, s] = np.log(np.std(y)) + 0.1 * rng.standard_normal()
        o = D + 1
        hyp[o, s] = logsn + 0.1 * rng.standard_normal()
        o += 1
        if noisefun[1] == 2:
            hyp[o, s] = 0.3 * rng.standard_normal()
            o += 1
        if noisefun[2] == 1:
            hyp[o, s] = np.median(y) + 0.1 * rng.standard_normal()      # threshold: about half of the outputs below it
            hyp[o + 1, s] = np.log(0.3) + 0.1 * rng.standard_normal()
            o += 2
        if meanfun >= 1:
            hyp[o, s] = np.max(y) + 0.1 * rng.standard_normal()
        if meanfun == 4:
            hyp[o + 1 : o + 1 + D, s] = 0.2 * rng.standard_normal(D)
            hyp[o + 1 + D :, s] = np.log(2.0) + 0.1 * rng.standard_normal(D)
    return dict(seed=seed, D=D, N=N, S=S, meanfun=meanfun, noisefun=list(noisefun), X=X, y=y, s2=s2, hyp=hyp, Xstar=Xstar,
                ystar=ystar, s2star=s2star)


def run_pred_case(c):
    N, D = c["X"].shape
    X = [[M(c["X"][n, d]) for d in range(D)] for n in range(N)]
    y = [M(t) for t in c["y"]]
    s2 = None if c["s2"] is None else [M(t) for t in c["s2"]]
    Xs = [[M(c["Xstar"][i, d]) for d in range(D)] for i in range(c["Xstar"].shape[0])]
    ystar = [M(t) for t in c["ystar"]]
    s2star = None if c["s2star"] is None else [M(t) for t in c["s2star"]]
    out = {k: [] for k in ("alpha", "fmu", "fs2", "ys2", "lp", "min_sn2")}
    for s in range(c["S"]):
        hyp = [M(t) for t in c["hyp"][:, s]]
        alpha, fmu, fs2, ys2, lp, mn = mp_pred_general(hyp, X, y, s2, Xs, ystar, s2star, c["meanfun"], c["noisefun"])
        for k, v in zip(("alpha", "fmu", "fs2", "ys2", "lp", "min_sn2"), (alpha, fmu, fs2, ys2, lp, mn)):
            out[k].append(fl(v))
    return out


def main_pred():
    outdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
    for i, spec in enumerate(PRED_CASES):
        c = make_pred_case(**spec)
        out = run_pred_case(c)
        rec = {"generator": "oracle/mp_golden.py pred (mpmath %s, dps=%d)" % (mp.__version__, mp.mp.dps),
               "inputs": {k: (tolist(v) if isinstance(v, np.ndarray) else v) for k, v in c.items()},
               "expected": out}
        path = os.path.join(outdir, "mp_pred_case%d.json" % i)
        with open(path, "w") as f:
            json.dump(rec, f)
        print("wrote", path, os.path.getsize(path), "bytes", file=sys.stderr)



# ------------------------------------------------------------------ soft-bound and weight penalties
def mp_penalties(theta, D, K, opt, fixed, lb, ub, TolCon, Wthresh, Wpen):
    """misc/vpbndloss.m:1-73 + utils/softbndloss.m:1-30 + the weight penalty of misc/negelcbo_vbmc.m:146-162, from the formulas:
    theta in the reference order [mu(:); ln sigma; ln lambda; eta] (optimised groups only), `fixed` the values of the others.
    Returns (L_bnd, dL_bnd[T], L_w, dL_w[T])."""
    i = 0
    if opt[0]:
        mu = theta[i : i + D * K]; i += D * K
    else:
        mu = fixed["mu"]
    if opt[1]:
        lns = theta[i : i + K]; i += K
    else:
        lns = [mp.log(t) for t in fixed["sigma"]]
    if opt[2]:
        lnl = theta[i : i + D]; i += D
    else:
        lnl = [mp.log(t) for t in fixed["lambda"]]
    eta = theta[i : i + K] if opt[3] else None
    ext, kind = [], []
    if opt[0]:
        ext += list(mu); kind += [("mu", p) for p in range(D * K)]
    if opt[1] or opt[2]:
        for k in range(K):                      # lnscale(:) of a D x K matrix: d fastest
            for d in range(D):
                ext.append(lns[k] + lnl[d]); kind.append(("sc", d, k))
    if opt[3]:
        ext += list(eta); kind += [("eta", k) for k in range(K)]
    assert len(ext) == len(lb) == len(ub)
    L = mp.mpf(0)
    dext = [mp.mpf(0)] * len(ext)
    for q, x in enumerate(ext):
        ell = (ub[q] - lb[q]) * TolCon
        if x < lb[q]:
            L += ((lb[q] - x) / ell) ** 2 / 2
            dext[q] = (x - lb[q]) / ell ** 2
        if x > ub[q]:
            L += ((x - ub[q]) / ell) ** 2 / 2
            dext[q] = (x - ub[q]) / ell ** 2
    T = len(theta)
    dL = [mp.mpf(0)] * T
    i = 0
    off = {}
    if opt[0]:
        off["mu"] = i; i += D * K
    if opt[1]:
        off["sigma"] = i; i += K
    if opt[2]:
        off["lambda"] = i; i += D
    if opt[3]:
        off["eta"] = i
    for q, kd in enumerate(kind):
        if kd[0] == "mu":
            dL[off["mu"] + kd[1]] += dext[q]
        elif kd[0] == "sc":
            if opt[1]:
                dL[off["sigma"] + kd[2]] += dext[q]
            if opt[2]:
                dL[off["lambda"] + kd[1]] += dext[q]
        else:
            dL[off["eta"] + kd[1]] += dext[q]
    Lw, dLw = mp.mpf(0), [mp.mpf(0)] * T
    if opt[3]:
        ee = [mp.e ** t for t in eta]
        ssum = mp.fsum(ee)
        w = [t / ssum for t in ee]
        Lw = Wpen * mp.fsum(wk if wk < Wthresh else Wthresh for wk in w)
        g = [Wpen if wk < Wthresh else mp.mpf(0) for wk in w]
        Jw = softmax_jac(eta)
        gw = matvec(Jw, g)
        for k in range(K):
            dLw[off["eta"] + k] = gw[k]
    return L, dL, Lw, dLw


PEN_CASES = [
    dict(seed=41, D=3, K=4, opt=(1, 1, 1, 1)),
    dict(seed=42, D=2, K=5, opt=(1, 1, 1, 0)),      # warm-up: weights fixed (setupvars_vbmc.m:89-91)
    dict(seed=43, D=4, K=3, opt=(1, 0, 0, 1)),
    dict(seed=44, D=2, K=6, opt=(0, 1, 1, 1)),
    dict(seed=45, D=3, K=2, opt=(1, 1, 0, 1)),
]


def make_pen_case(seed, D, K, opt):
    rng = np.random.default_rng(seed)
    mu = 1.5 * rng.standard_normal((D, K))
    sigma = 0.4 * np.exp(0.3 * rng.standard_normal(K))
    lam = np.exp(0.2 * rng.standard_normal(D))
    eta = 1.5 * rng.standard_normal(K)
    eta[0] = -6.0                                   # a weight below the threshold
    theta = np.concatenate([x for x, o in ((mu.reshape(-1, order="F"), opt[0]), (np.log(sigma), opt[1]), (np.log(lam), opt[2]), (eta, opt[3])) if o])
    next_ = (D * K if opt[0] else 0) + (D * K if (opt[1] or opt[2]) else 0) + (K if opt[3] else 0)
    # bounds that some entries violate on either side (and some sit exactly on)
    lb, ub = [], []
    if opt[0]:
        lb += list(mu.reshape(-1, order="F") - np.where(rng.random(D * K) < 0.3, -0.4, 1.0))
        ub += list(mu.reshape(-1, order="F") + np.where(rng.random(D * K) < 0.3, -0.3, 1.2))
    if opt[1] or opt[2]:
        sc = (np.log(sigma)[None, :] + np.log(lam)[:, None]).reshape(-1, order="F")
        lb += list(sc - np.where(rng.random(D * K) < 0.25, -0.2, 0.8))
        ub += list(sc + np.where(rng.random(D * K) < 0.25, -0.25, 0.9))
    if opt[3]:
        lb += list(np.full(K, -4.0))
        ub += list(np.zeros(K))
    lb, ub = np.array(lb), np.array(ub)
    swap = lb > ub                                  # keep lb < ub where both shifts went inwards
    lb[swap], ub[swap] = ub[swap] - 0.1, lb[swap] + 0.1
    assert lb.size == next_
    return dict(seed=seed, D=D, K=K, opt=list(opt), theta=theta, mu=mu, sigma=sigma, lam=lam, eta=eta, lb=lb, ub=ub,
                TolCon=0.01, WeightThreshold=0.05, WeightPenalty=0.1)


def run_pen_case(c):
    D, K, opt = c["D"], c["K"], c["opt"]
    fixed = {"mu": [M(t) for t in c["mu"].reshape(-1, order="F")], "sigma": [M(t) for t in c["sigma"]], "lambda": [M(t) for t in c["lam"]]}
    L, dL, Lw, dLw = mp_penalties([M(t) for t in c["theta"]], D, K, opt, fixed, [M(t) for t in c["lb"]], [M(t) for t in c["ub"]],
                                  M(c["TolCon"]), M(c["WeightThreshold"]), M(c["WeightPenalty"]))
    return {"L_bnd": fl(L), "dL_bnd": fl(dL), "L_w": fl(Lw), "dL_w": fl(dLw)}


def main_pen():
    outdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
    for i, spec in enumerate(PEN_CASES):
        c = make_pen_case(**spec)
        out = run_pen_case(c)
        rec = {"generator": "oracle/mp_golden.py pen (mpmath %s, dps=%d)" % (mp.__version__, mp.mp.dps),
               "inputs": {k: (tolist(v) if isinstance(v, np.ndarray) else v) for k, v in c.items()},
               "expected": out}
        path = os.path.join(outdir, "mp_pen_case%d.json" % i)
        with open(path, "w") as f:
            json.dump(rec, f)
        print("wrote", path, os.path.getsize(path), "bytes", file=sys.stderr)


# ------------------------------------------------------------------ driver
def tolist(a):
    return np.asarray(a, dtype=np.float64).tolist()


def fl(x):
    if isinstance(x, (list, tuple)):
        return [fl(t) for t in x]
    return float(x)


def make_case(seed, D, K, N, S, Mh, meanfun, target="quad"):
    rng = np.random.default_rng(seed)
    X = 1.5 * rng.standard_normal((N, D))
    if target == "rosenbrock":
        # BASELINE configs[0]: the reference's own test target, rosenbrock_test.m:7 (noise-free form):
        #   y = -sum((x(:,1:end-1).^2 - x(:,2:end)).^2 + (x(:,1:end-1) - 1).^2/100, 2)
        y = -np.sum((X[:, :-1] ** 2 - X[:, 1:]) ** 2 + (X[:, :-1] - 1.0) ** 2 / 100.0, axis=1)
    else:
        y = -0.5 * np.sum((X / 1.3) ** 2, axis=1) + 0.3 * np.sin(X[:, 0]) + 0.05 * rng.standard_normal(N)
    nmean = {0: 0, 1: 1, 4: 2 * D + 1}[meanfun]
    hyp = np.zeros((D + 2 + nmean, S))
    for s in range(S):
        hyp[:D, s] = np.log(0.8) + 0.2 * rng.standard_normal(D)
        hyp[D, s] = np.log(np.std(y)) + 0.1 * rng.standard_normal()
        hyp[D + 1, s] = np.log(3e-2) + 0.1 * rng.standard_normal()
        if meanfun >= 1:
            hyp[D + 2, s] = np.max(y) + 0.1 * rng.standard_normal()
        if meanfun == 4:
            hyp[D + 3 : D + 3 + D, s] = 0.2 * rng.standard_normal(D)
            hyp[D + 3 + D :, s] = np.log(2.0) + 0.1 * rng.standard_normal(D)
    mu = X[rng.permutation(N)[:K]].T + 0.1 * rng.standard_normal((D, K))
    sigma = 0.4 * np.exp(0.3 * rng.standard_normal(K))
    lam = np.exp(0.2 * rng.standard_normal(D))
    lam = lam / np.sqrt(np.sum(lam**2) / D)
    eta = 0.5 * rng.standard_normal(K)
    w = np.exp(eta) / np.sum(np.exp(eta))
    eps = rng.standard_normal((K, Mh, D))
    Xstar = 1.5 * rng.standard_normal((5, D))
    c = dict(seed=seed, D=D, K=K, N=N, S=S, Mh=Mh, meanfun=meanfun, X=X, y=y, hyp=hyp, mu=mu, sigma=sigma,
             lam=lam, eta=eta, w=w, eps=eps, Xstar=Xstar)
    if target != "quad":
        c["target"] = target
    return c


def run_case(c):
    D, K, N, S = c["D"], c["K"], c["N"], c["S"]
    mu = [[M(c["mu"][d, k]) for k in range(K)] for d in range(D)]
    sigma = [M(t) for t in c["sigma"]]
    lam = [M(t) for t in c["lam"]]
    eta = [M(t) for t in c["eta"]]
    # weights exactly as negelcbo_vbmc.m:45-47 forms them from eta
    ee = [mp.e ** t for t in eta]
    w = [t / mp.fsum(ee) for t in ee]
    eps = [[[M(c["eps"][j, i, d]) for d in range(D)] for i in range(c["Mh"])] for j in range(K)]
    X = [[M(c["X"][n, d]) for d in range(D)] for n in range(N)]
    y = [M(t) for t in c["y"]]
    out = {}
    H, dH = mp_entmc(mu, sigma, lam, w, eta, eps)
    out["entmc_H"], out["entmc_dH"] = fl(H), fl(dH)
    Hl, dHl = mp_entlb(mu, sigma, lam, w, eta)
    out["entlb_H"], out["entlb_dH"] = fl(Hl), fl(dHl)
    posts = []
    out["alpha"], out["L"], out["pred_fmu"], out["pred_fs2"] = [], [], [], []
    Xs = [[M(c["Xstar"][i, d]) for d in range(D)] for i in range(c["Xstar"].shape[0])]
    for s in range(S):
        hyp = [M(t) for t in c["hyp"][:, s]]
        alpha, L, sn2 = mp_gp_post(hyp, X, y, c["meanfun"])
        posts.append(dict(hyp=hyp, alpha=alpha, L=L, sn2=sn2))
        out["alpha"].append(fl(alpha))
        out["L"].append(fl(L))
        fmu, fs2 = mp_gp_pred(hyp, X, alpha, L, sn2, Xs, c["meanfun"])
        out["pred_fmu"].append(fl(fmu))
        out["pred_fs2"].append(fl(fs2))
    Fs, dFs, I_sk, J, varF1 = mp_logjoint(mu, sigma, lam, w, eta, X, posts, c["meanfun"], 1)
    _, _, _, _, varF2 = mp_logjoint(mu, sigma, lam, w, eta, X, posts, c["meanfun"], 2)
    out["G_s"], out["dG_s"], out["I_sk"], out["J_sjk"] = fl(Fs), fl(dFs), fl(I_sk), fl(J)
    out["varG_s_full"], out["varG_s_diag"] = fl(varF1), fl(varF2)
    return out


CASES = [
    dict(seed=11, D=2, K=2, N=6, S=1, Mh=8, meanfun=4),
    dict(seed=12, D=3, K=4, N=12, S=3, Mh=6, meanfun=4),
    dict(seed=13, D=1, K=3, N=8, S=2, Mh=10, meanfun=1),
    dict(seed=14, D=4, K=3, N=10, S=2, Mh=4, meanfun=0),
    # BASELINE configs[0]: the Rosenbrock target of the reference's rosenbrock_test.m as the GP's training data, D = 2, K = 2
    # mixture, Ns = 100 (Mh = 50 antithetic pairs), one hyper-parameter sample
    dict(seed=15, D=2, K=2, N=20, S=1, Mh=50, meanfun=4, target="rosenbrock"),
]


# ------------------------------------------------------------------ acquisition functions
def mp_lse(z):
    m = max(z)
    return m + mp.log(mp.fsum(mp.e ** (t - m) for t in z))


def run_acq_case(c):
    """acqwrapper_vbmc.m:17-33 without the variance regulariser, for acqf / acqflog / acqus / acqfsn2 / acqviqr
    (acq/acqf_vbmc.m:6-10, acqflog_vbmc.m:14-18, acqus_vbmc.m:6-9, acqfsn2_vbmc.m:6-17, acqviqr_vbmc.m:36-109), from the
    definitions: exact GP algebra and exact log-sum-exps in 50 digits."""
    D, K, N, S = c["D"], c["K"], c["N"], c["S"]
    X = [[M(c["X"][n, d]) for d in range(D)] for n in range(N)]
    y = [M(t) for t in c["y"]]
    Xs = [[M(c["Xstar"][i, d]) for d in range(D)] for i in range(c["Xstar"].shape[0])]
    Xa = [[M(c["Xa"][i, d]) for d in range(D)] for i in range(c["Xa"].shape[0])]
    Nx, Na = len(Xs), len(Xa)
    mu = [[M(c["mu"][d, k]) for k in range(K)] for d in range(D)]
    sigma = [M(t) for t in c["sigma"]]
    lam = [M(t) for t in c["lam"]]
    w = [M(t) for t in c["w"]]
    gl = [M(t) for t in c["gplengthscale"]]
    sn2new = [M(t) for t in c["sn2new"]]
    ymax = M(c["ymax"])
    u = M(0.6745)
    fmu, fs2, acq_s = [], [], []
    for s in range(S):
        hyp = [M(t) for t in c["hyp"][:, s]]
        alpha, L, sn2 = mp_gp_post(hyp, X, y, c["meanfun"])
        a, b = mp_gp_pred(hyp, X, alpha, L, sn2, Xs, c["meanfun"])
        fmu.append(a)
        fs2.append(b)
    fbar = [mp.fsum(fmu[s][i] for s in range(S)) / S for i in range(Nx)]
    vbar = [mp.fsum(fs2[s][i] for s in range(S)) / S for i in range(Nx)]
    vf = [mp.fsum((fmu[s][i] - fbar[i]) ** 2 for s in range(S)) / (S - 1) if S > 1 else mp.mpf(0) for i in range(Nx)]
    vtot = [vf[i] + vbar[i] for i in range(Nx)]
    # vbmc_pdf in the transformed space (vbmc_pdf.m:86-97)
    pdf = []
    for x in Xs:
        acc = mp.mpf(0)
        for k in range(K):
            q = mp.fsum(((x[d] - mu[d][k]) / (sigma[k] * lam[d])) ** 2 for d in range(D))
            nf = (2 * mp.pi) ** (-mp.mpf(D) / 2) / mp.fprod(sigma[k] * lam[d] for d in range(D))
            acc += w[k] * nf * mp.e ** (-q / 2)
        pdf.append(acc)
    # nearest training input in the rescaled space (acqfsn2_vbmc.m:10-12, acqviqr_vbmc.m:48-50)
    pos = []
    for x in Xs:
        d2 = [mp.fsum((x[d] / gl[d] - X[n][d] / gl[d]) ** 2 for d in range(D)) for n in range(N)]
        pos.append(d2.index(min(d2)))
    out = {
        "fbar": fl(fbar), "vtot": fl(vtot),
        "acqf": fl([-vtot[i] * mp.e ** (fbar[i] - ymax) * pdf[i] for i in range(Nx)]),
        "acqflog": fl([-(mp.log(vtot[i]) + fbar[i] - ymax + mp.log(pdf[i])) for i in range(Nx)]),
        "acqus": fl([-vtot[i] * pdf[i] ** 2 for i in range(Nx)]),
        "acqfsn2": fl([-vtot[i] * (1 - sn2new[pos[i]] / (vtot[i] + sn2new[pos[i]])) * mp.e ** (fbar[i] - ymax) * pdf[i] for i in range(Nx)]),
        "pos": pos,
    }
    # VIQR: expected log IQR of the posterior after observing at x, importance points Xa with unit weights
    for s in range(S):
        hyp = [M(t) for t in c["hyp"][:, s]]
        ell = [mp.e ** hyp[d] for d in range(D)]
        sf2 = mp.e ** (2 * hyp[D])
        alpha, L, sn2 = mp_gp_post(hyp, X, y, c["meanfun"])
        kern = lambda p, q: sf2 * mp.e ** (-mp.fsum(((p[d] - q[d]) / ell[d]) ** 2 for d in range(D)) / 2)  # noqa: E731
        _, fs2a = mp_gp_pred(hyp, X, alpha, L, sn2, Xa, c["meanfun"])
        # Ctmp(:, a) = (L \ (L' \ k(X, xa))) / sn2_eff   (activeimportancesampling_vbmc.m:271)
        Ct = [[t / sn2 for t in mp_solve_ut(L, mp_solve_ut_t(L, [kern(X[n], xa) for n in range(N)]))] for xa in Xa]
        row = []
        for i, x in enumerate(Xs):
            ks = [kern(X[n], x) for n in range(N)]
            ys2 = fs2[s][i] + sn2new[pos[i]]
            zz = []
            for a_, xa in enumerate(Xa):
                C = kern(x, xa) - mp.fsum(ks[n] * Ct[a_][n] for n in range(N))          # acqviqr_vbmc.m:84-86
                sp = mp.sqrt(max(fs2a[a_] - C * C / ys2, mp.mpf(0)))                      # :93-95
                zz.append(u * sp + mp.log1p(-mp.e ** (-2 * u * sp)))                      # :97-99
            row.append(mp_lse(zz))
        acq_s.append(row)
    out["acqviqr"] = fl([mp_lse([acq_s[s][i] for s in range(S)]) - mp.log(S) for i in range(Nx)])   # :107-109
    return out


ACQ_CASES = [
    dict(seed=21, D=2, K=3, N=9, S=1, Mh=2, meanfun=4),
    dict(seed=22, D=3, K=4, N=12, S=3, Mh=2, meanfun=4),
    dict(seed=23, D=4, K=2, N=10, S=2, Mh=2, meanfun=1),
]


def main_acq():
    outdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
    for i, spec in enumerate(ACQ_CASES):
        c = make_case(**spec)
        rng = np.random.default_rng(1000 + spec["seed"])
        c["Xa"] = 1.2 * rng.standard_normal((6, spec["D"]))
        c["gplengthscale"] = np.exp(0.2 * rng.standard_normal(spec["D"]))
        c["sn2new"] = 1e-3 * (1.0 + rng.random(spec["N"]))
        c["ymax"] = float(np.max(c["y"]))
        out = run_acq_case(c)
        rec = {"generator": "oracle/mp_golden.py acq (mpmath %s, dps=%d)" % (mp.__version__, mp.mp.dps),
               "inputs": {k: (tolist(v) if isinstance(v, np.ndarray) else v) for k, v in c.items()},
               "expected": out}
        path = os.path.join(outdir, "mp_acq_case%d.json" % i)
        with open(path, "w") as f:
            json.dump(rec, f)
        print("wrote", path, os.path.getsize(path), "bytes", file=sys.stderr)


def main(only=None):
    outdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    for i, spec in enumerate(CASES):
        if only is not None and i not in only:
            continue
        c = make_case(**spec)
        out = run_case(c)
        rec = {"generator": "oracle/mp_golden.py (mpmath %s, dps=%d)" % (mp.__version__, mp.mp.dps),
               "inputs": {k: (tolist(v) if isinstance(v, np.ndarray) else v) for k, v in c.items()},
               "expected": out}
        path = os.path.join(outdir, "mp_case%d.json" % i)
        with open(path, "w") as f:
            json.dump(rec, f)
        print("wrote", path, os.path.getsize(path), "bytes", file=sys.stderr)


def main_nlz():
    outdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
    for i, spec in enumerate(NLZ_CASES):
        c = make_nlz_case(**spec)
        out = run_nlz_case(c)
        rec = {"generator": "oracle/mp_golden.py nlz (mpmath %s, dps=%d)" % (mp.__version__, mp.mp.dps),
               "inputs": {k: (tolist(v) if isinstance(v, np.ndarray) else v) for k, v in c.items()},
               "expected": out}
        path = os.path.join(outdir, "mp_nlz_case%d.json" % i)
        with open(path, "w") as f:
            json.dump(rec, f)
        print("wrote", path, os.path.getsize(path), "bytes", file=sys.stderr)


if __name__ == "__main__":
    if "rosenbrock" in sys.argv[1:]:
        main(only=[i for i, c in enumerate(CASES) if c.get("target") == "rosenbrock"])   # only mp_case4.json (BASELINE configs[0])
    elif "nlz" in sys.argv[1:]:
        main_nlz()      # only the marginal-likelihood fixtures
    elif "pred" in sys.argv[1:]:
        main_pred()
    elif "pen" in sys.argv[1:]:
        main_pen()
    elif "acq" in sys.argv[1:]:
        main_acq()      # only the acquisition-function fixtures
    else:
        main()
        main_nlz()
        main_acq()
        main_pred()
        main_pen()
