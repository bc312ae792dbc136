"""The callers of the ELBO objective: sieve, Adam, vpoptimize (host orchestration).

Mirrors misc/vpsieve_vbmc.m, misc/vbinit_vbmc.m, misc/gethpd_vbmc.m, utils/fminadam.m and the
Adam path of misc/vpoptimize_vbmc.m (+ eval_fullelcbo) with the same names and argument meaning.
What changes is the execution shape: the sieve's R sequential negelcbo calls (vpsieve_vbmc.m:74-78)
become ONE batched device pass, optionally sharded over ranks (vbmc_amd/dist.py), and the Nslowopts
Adam chains advance in lock-step, one batched ELBO+grad evaluation per iteration.

Random draws (vbinit jitter, MC entropy) come from a numpy Generator / the device Philox stream:
MATLAB's global randn stream cannot be reproduced, so these functions are structurally, not
bit-wise, identical to a MATLAB run; the numerics of every objective evaluation are parity-tested.
"""
from __future__ import annotations

import math

import numpy as np

from .elbo import fminadam_device, negelcbo_batch
from .vp import DEFAULT_OPTIONS, copy_vp, evaloption, get_vptheta, rescale_params, vpbounds

EPS = float(np.finfo(np.float64).eps)


def _mround(x):
    return int(math.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)


def sort_ascend(v):
    """MATLAB sort(v,'ascend') indices: stable, NaN last (vpsieve_vbmc.m:81)."""
    v = np.asarray(v, dtype=np.float64)
    nan = np.isnan(v)
    good = np.nonzero(~nan)[0]
    return np.concatenate([good[np.argsort(v[good], kind="stable")], np.nonzero(nan)[0]])


def sort_descend(v):
    v = np.asarray(v, dtype=np.float64)
    nan = np.isnan(v)
    idx = np.argsort(-np.where(nan, 0.0, v), kind="stable")
    return np.concatenate([np.nonzero(nan)[0], idx[~nan[idx]]])


def gethpd_vbmc(X, y, HPDFrac=0.8):
    """misc/gethpd_vbmc.m:9-14."""
    order = sort_descend(y)
    n = _mround(HPDFrac * X.shape[0])
    return X[order[:n], :], y[order[:n]]


def vbinit_vbmc(type_, Nopts, vp, Knew, Xstar, ystar, rng):
    """misc/vbinit_vbmc.m:1-140 -> (vp0_vec, type_vec)."""
    D, K = vp["D"], vp["K"]
    Nstar = Xstar.shape[0]
    lambda0, mu0, w0 = vp["lambda"].copy(), vp["mu"].copy(), vp["w"].copy()
    if type_ == 1:
        sigma0 = vp["sigma"].copy()
    elif type_ == 2:
        order = sort_descend(ystar)
        if vp["optimize_mu"]:
            idx_ord = np.tile(np.arange(min(Knew, Nstar)), int(math.ceil(Knew / Nstar)))
            mu0 = Xstar[order[idx_ord[:Knew]], :].T.copy()
        V = np.var(mu0, axis=1, ddof=1) if K > 1 else np.var(Xstar, axis=0, ddof=1)
        sigma0 = np.sqrt(np.mean(V / lambda0**2) / Knew) * np.exp(0.2 * rng.standard_normal(Knew))
    elif type_ == 3:
        if vp["optimize_mu"]:
            mu0 = np.zeros((D, K))
        sigma0 = np.zeros(K)
    else:
        raise ValueError("vbinit:UnknownType Unknown TYPE for initialization of variational posteriors.")
    out = []
    for iOpt in range(Nopts):
        v = copy_vp(vp)
        v["K"] = Knew
        mu, sigma, lam = mu0.copy(), sigma0.copy(), lambda0.copy()
        w = w0.copy() if vp["optimize_weights"] else None
        add_jitter = True
        if type_ == 1:
            if iOpt == 0:
                add_jitter = False
            if Knew > K:
                mu = np.concatenate([mu, np.zeros((D, Knew - K))], axis=1)
                sigma = np.concatenate([sigma, np.zeros(Knew - K)])
                if w is not None:
                    w = np.concatenate([w, np.zeros(Knew - K)])
                for iNew in range(K, Knew):
                    idx = int(rng.integers(K))
                    mu[:, iNew] = mu[:, idx]
                    sigma[iNew] = sigma[idx]
                    mu[:, iNew] += 0.5 * sigma[iNew] * lam * rng.standard_normal(D)
                    if vp["optimize_sigma"]:
                        sigma[iNew] *= math.exp(0.2 * rng.standard_normal())
                    if vp["optimize_weights"]:
                        xi = 0.25 + 0.25 * rng.random()
                        w[iNew] = xi * w[idx]
                        w[idx] = (1 - xi) * w[idx]
        elif type_ == 2:
            if iOpt == 0:
                add_jitter = False
            if vp["optimize_lambda"]:
                lam = np.std(Xstar, axis=0, ddof=1)
                lam = lam * math.sqrt(D / np.sum(lam**2))
            if vp["optimize_weights"]:
                w = np.ones(Knew) / Knew
        else:
            order = rng.permutation(Nstar)
            if vp["optimize_mu"]:
                idx_ord = np.tile(np.arange(min(Knew, Nstar)), int(math.ceil(Knew / Nstar)))
                mu = Xstar[order[idx_ord[:Knew]], :].T.copy()
            V = np.var(mu, axis=1, ddof=1) if K > 1 else np.var(Xstar, axis=0, ddof=1)
            if vp["optimize_sigma"]:
                sigma = math.sqrt(np.mean(V) / Knew) * np.exp(0.2 * rng.standard_normal(Knew))
            if vp["optimize_lambda"]:
                lam = np.std(Xstar, axis=0, ddof=1)
                lam = lam * math.sqrt(D / np.sum(lam**2))
            if vp["optimize_weights"]:
                w = np.ones(Knew) / Knew
        if add_jitter:
            if vp["optimize_mu"]:
                mu = mu + sigma[None, :] * lam[:, None] * rng.standard_normal(mu.shape)
            if vp["optimize_sigma"]:
                sigma = sigma * np.exp(0.2 * rng.standard_normal(Knew))
            if vp["optimize_lambda"]:
                lam = lam * np.exp(0.2 * rng.standard_normal(D))
            if vp["optimize_weights"]:
                w = w * np.exp(0.2 * rng.standard_normal(Knew))
                w = w / np.sum(w)
        v["w"] = w if vp["optimize_weights"] else np.ones(Knew) / Knew
        v["mu"] = mu if vp["optimize_mu"] else mu0
        v["sigma"] = sigma
        v["lambda"] = lam
        out.append(v)
    return out, np.full(Nopts, type_, dtype=np.int64)


def sieve_evaluate(vp0_vec, gp, NSentKFast, compute_var, elcbo_beta, thetabnd, *, seed=0, engine=None, shard=None):
    """The batched form of vpsieve_vbmc.m:74-78: nelcbo_fill(i) = nelbo + beta*sqrt(varF).

    ``shard`` = (rank, world, allgather) evaluates only candidates i = rank (mod world) and
    all-gathers the values (vbmc_amd/dist.py); every rank returns the full vector.
    """
    thetas, vps = [], []
    for v in vp0_vec:
        th, v2 = get_vptheta(v)
        thetas.append(th)
        vps.append(v2)
    Th = np.asfortranarray(np.stack(thetas, axis=1))
    R = Th.shape[1]
    idx = np.arange(R)
    if shard is not None:
        rank, world, allgather = shard
        idx = idx[rank::world]
    vals = np.full(R, np.nan)
    # the batched ABI shares the NON-optimised parameter groups across the batch (they are not part of theta);
    # candidates that differ in such a group (e.g. fixed weights) are evaluated in separate sub-batches
    fixed = [g for g, f in (("mu", "optimize_mu"), ("sigma", "optimize_sigma"), ("lambda", "optimize_lambda"),
                            ("w", "optimize_weights")) if not vps[0][f]]
    groups = {}
    for i in idx:
        key = tuple(np.asarray(vps[i][g], dtype=np.float64).tobytes() for g in fixed)
        groups.setdefault(key, []).append(int(i))
    local = np.zeros(idx.size)
    pos = {int(i): p for p, i in enumerate(idx)}
    for members in groups.values():
        out = negelcbo_batch(Th[:, members], 0, vps[members[0]], gp, NSentKFast, False, int(bool(compute_var)), thetabnd,
                             seed=seed, engine=engine)
        v = out["F"] + elcbo_beta * np.sqrt(out["varG"]) if compute_var else out["F"]
        local[[pos[i] for i in members]] = v
    if shard is not None:
        vals = allgather(local, idx, R)
    else:
        vals[idx] = local
    return vals, vps


def vpsieve_vbmc(Ninit, Nbest, vp, gp, optimState=None, options=None, K=None, *, rng=None, seed=0, engine=None, shard=None):
    """[vp0_vec,vp0_type,elcbo_beta,compute_var,NSentK,NSentKFast] = vpsieve_vbmc(...)  (misc/vpsieve_vbmc.m:1-90)."""
    options = dict(DEFAULT_OPTIONS, **(options or {}))
    optimState = dict(optimState or {})
    rng = np.random.default_rng(0) if rng is None else rng
    optimState.setdefault("delta", 0)
    optimState.setdefault("EntropySwitch", False)
    optimState.setdefault("Neff", gp["X"].shape[0])
    if Nbest is None:
        Nbest = 1
    K = vp["K"] if K is None else K
    vp = copy_vp(vp)
    vp["delta"] = optimState["delta"]
    if Ninit is None:
        Ninit = int(math.ceil(evaloption(options["NSelbo"], K)))
    NSentK = int(math.ceil(evaloption(options["NSent"], K) / K))
    NSentKFast = int(math.ceil(evaloption(options["NSentFast"], K) / K))
    if optimState["EntropySwitch"] or K == 1:
        NSentK = NSentKFast = 0  # :30-33
    elcbo_beta = evaloption(options["ELCBOWeight"], optimState["Neff"])
    compute_var = elcbo_beta != 0
    vp, thetabnd = vpbounds(vp, gp, options, K)
    if Ninit > 0:
        Xstar, ystar = gethpd_vbmc(gp["X"], gp["y"], options["HPDFrac"])
        if Nbest == 1:
            vp0_vec, vp0_type = vbinit_vbmc(1, Ninit, vp, K, Xstar, ystar, rng)
        else:
            n3 = int(math.ceil(Ninit / 3))
            a, ta = vbinit_vbmc(1, n3, vp, K, Xstar, ystar, rng)
            b, tb = vbinit_vbmc(2, n3, vp, K, Xstar, ystar, rng)
            c, tc = vbinit_vbmc(3, Ninit - 2 * n3, vp, K, Xstar, ystar, rng)
            vp0_vec, vp0_type = a + b + c, np.concatenate([ta, tb, tc])
        repo = optimState.get("vp_repo")
        if repo and options.get("VariationalInitRepo"):     # :62-72: earlier solutions with the same number of parameters
            Ntheta = get_vptheta(vp0_vec[0])[0].size
            extra = [rescale_params(vp0_vec[0], th) for th in repo if np.size(th) == Ntheta]
            vp0_vec = vp0_vec + extra
            vp0_type = np.concatenate([vp0_type, np.ones(len(extra), dtype=np.int64)])
        nelcbo_fill, vp0_vec = sieve_evaluate(vp0_vec, gp, NSentKFast, compute_var, elcbo_beta, thetabnd, seed=seed,
                                              engine=engine, shard=shard)
        order = sort_ascend(nelcbo_fill)  # :81
        vp0_vec = [vp0_vec[i] for i in order]
        vp0_type = vp0_type[order]
    else:
        vp0_vec, vp0_type, nelcbo_fill = [vp], np.array([1]), np.zeros(0)
    return vp0_vec, vp0_type, elcbo_beta, compute_var, NSentK, NSentKFast, nelcbo_fill


def _polyfit1(x, y):
    n = x.size
    xm = np.mean(x)
    sxx = np.sum((x - xm) ** 2)
    slope = np.sum((x - xm) * (y - np.mean(y))) / sxx
    icpt = np.mean(y) - slope * xm
    r = y - (slope * x + icpt)
    return float(slope), float(np.sum(r * r)) / (n - 2) / sxx


def fminadam(fun, x0, LB=None, UB=None, TolFun=0.001, MaxIter=10000, master_stepsize=None):
    """[x,f,xtab,ftab,iter] = fminadam(fun,x0,LB,UB,TolFun,MaxIter,master_stepsize)  (utils/fminadam.m:1-104)."""
    ms = {"max": 0.1, "min": 0.001, "decay": 200.0}
    if master_stepsize:
        ms.update({k: v for k, v in master_stepsize.items() if v is not None})
    fudge = math.sqrt(EPS)
    b1, b2, batch = 0.9, 0.999, 20
    TolX, TolX_max, TolFun_max = 0.001, 0.1, TolFun * 100.0
    x0 = np.asarray(x0, dtype=np.float64)
    nv = x0.size
    LB = np.full(nv, -np.inf) if LB is None else np.asarray(LB, dtype=np.float64).reshape(-1)
    UB = np.full(nv, np.inf) if UB is None else np.asarray(UB, dtype=np.float64).reshape(-1)
    m = np.zeros(nv)
    v = np.zeros(nv)
    MaxIter = int(MaxIter)
    xtab = np.zeros((nv, MaxIter))
    ftab = np.full(MaxIter, np.nan)
    x = x0.reshape(-1).copy()
    it = 0
    for it in range(1, MaxIter + 1):
        f, g = fun(x)
        ftab[it - 1] = f
        g = np.asarray(g, dtype=np.float64).reshape(-1)
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g**2
        mhat = m / (1 - b1**it)
        vhat = v / (1 - b2**it)
        step = ms["min"] + (ms["max"] - ms["min"]) * math.exp(-it / ms["decay"])
        x = x - step * mhat / (np.sqrt(vhat) + fudge)
        x = np.minimum(np.maximum(x, LB), UB)
        xtab[:, it - 1] = x
        if it % batch == 0 and it >= 2 * batch:
            xxp = np.linspace(-(batch - 1) / 2.0, (batch - 1) / 2.0, batch)
            slope, svar = _polyfit1(xxp, ftab[it - batch:it])
            slope_err = math.sqrt(svar + TolFun**2)
            slope_err_max = math.sqrt(svar + TolFun_max**2)
            dx = math.sqrt(np.sum((np.mean(xtab[:, it - batch:it], axis=1) - np.mean(xtab[:, it - 2 * batch:it - batch], axis=1)) ** 2 / batch))
            if (dx < TolX and abs(slope) < slope_err_max) or (abs(slope) < slope_err and dx < TolX_max):
                break
    xo = np.mean(xtab[:, it - batch:it], axis=1)
    fo = float(np.mean(ftab[it - batch:it]))
    return xo.reshape(x0.shape), fo, xtab[:, :it], ftab[:it], it


_FIXED_GROUPS = (("mu", "optimize_mu"), ("sigma", "optimize_sigma"), ("lambda", "optimize_lambda"), ("w", "optimize_weights"))


def _group_by_fixed(vps):
    """The batched ABI shares the NON-optimised parameter groups across a batch (they are not part of theta): indices of
    ``vps`` grouped by the bytes of those groups, in first-appearance order.  One group in every default VBMC run."""
    fixed = [g for g, f in _FIXED_GROUPS if not vps[0][f]]
    groups = {}
    for i, v in enumerate(vps):
        groups.setdefault(tuple(np.asarray(v[g], dtype=np.float64).tobytes() for g in fixed), []).append(i)
    return list(groups.values())


def eval_fullelcbo_batch(thetas, vps, gp, beta, options, *, seed=0, engine=None, trace=None, kind="full", labels=None):
    """eval_fullelcbo (misc/vpoptimize_vbmc.m:257-305) for several (theta, vp0) pairs in ONE device pass per group of
    equal fixed groups: NSentFine samples, full variance, per-component terms.  Slot i of the batch draws the device
    stream (seed, r = position in its group); ``trace`` records that for tests that replay the stream."""
    K = vps[0]["K"]
    NSentFineK = int(math.ceil(evaloption(options["NSentFine"], K) / K))
    cv = 0 if options.get("SkipELBOVariance") else 1
    stats = [None] * len(vps)
    for gi, members in enumerate(_group_by_fixed(vps)):
        Th = np.asfortranarray(np.stack([np.asarray(thetas[i], dtype=np.float64).reshape(-1) for i in members], axis=1))
        sd = seed + (gi << 12)
        out = negelcbo_batch(Th, 0, vps[members[0]], gp, NSentFineK, False, cv, None, separate_K=True, seed=sd, engine=engine)
        for r, i in enumerate(members):
            varF = float(out["varG"][r]) if cv else 0.0
            stats[i] = {"nelbo": float(out["F"][r]), "G": float(out["G"][r]), "H": float(out["H"][r]), "varF": varF, "varG": varF,
                        "varH": 0.0, "varss": float(out["varGss"][r]) if cv else 0.0,
                        "nelcbo": float(out["F"][r]) + beta * math.sqrt(varF), "theta": Th[:, r].copy(),
                        "I_sk": out["I_sk"][:, :, r].copy(), "J_sjk": out["J_sjk"][:, :, :, r].copy() if cv else None}
            if trace is not None:
                trace.append({"kind": kind, "slot": labels[i] if labels is not None else i, "seed": sd, "r": r, "R": len(members),
                              "K": K, "Ns": NSentFineK})
    return stats


def _stats_from_cache(I_sk, J_sjk, w):
    """G, varG, varss of a mixture from the per-component terms I_sk (S x K) and J_sjk (S x K x K, full variance) and ITS weights:
    misc/gplogjoint.m:204,329-332,350 (the sums over components) and :399-413 (the statistics over hyper-samples); cf. the
    reference's misc/gplogjoint_weights.m:47-58, which recombines the same cached arrays."""
    S = I_sk.shape[0]
    F = I_sk @ w
    d = np.einsum("skk->sk", J_sjk)
    varF = np.sum(w ** 2 * np.maximum(EPS, d), axis=1) + (np.einsum("sjk,j,k->s", J_sjk, w, w) - np.sum(w ** 2 * d, axis=1))
    varF = np.maximum(varF, EPS)
    varss = 0.0
    if S > 1:
        Fbar = float(np.sum(F) / S)
        varFss = float(np.sum((F - Fbar) ** 2) / (S - 1))
        varss = varFss + float(np.std(varF, ddof=1))
        G, varG = Fbar, float(np.sum(varF) / S + varFss)
    else:
        G, varG = float(F[0]), float(varF[0])
    return {"G": G, "varF": varG, "varG": varG, "varss": varss, "I_sk": I_sk.copy(), "J_sjk": J_sjk.copy()}


def eval_fullelcbo(theta, vp, gp, beta, options, *, seed=0, engine=None):
    """eval_fullelcbo (misc/vpoptimize_vbmc.m:257-305) for one theta."""
    return eval_fullelcbo_batch([theta], [vp], gp, beta, options, seed=seed, engine=engine)[0]


def fminadam_lockstep(fun_batch, X0, TolFun=0.001, MaxIter=10000, master_stepsize=None):
    """utils/fminadam.m:1-104 for R chains advanced in lock-step by ONE batched objective call per iteration
    (``fun_batch(X (T x R), it) -> (F (R), dF (T x R))``), every chain with its own stopping test (:65-81); a chain that
    has stopped is still evaluated (its column is ignored), exactly like vbmc_adam_batch on the device.
    Returns (x (T x R), f (R), xtab list, ftab list, iters (R))."""
    ms = {"max": 0.1, "min": 0.001, "decay": 200.0}
    if master_stepsize:
        ms.update({k: v for k, v in master_stepsize.items() if v is not None})
    fudge = math.sqrt(EPS)
    b1, b2, batch = 0.9, 0.999, 20
    TolX, TolX_max, TolFun_max = 0.001, 0.1, TolFun * 100.0
    X = np.array(X0, dtype=np.float64, order="F")
    T, R = X.shape
    MaxIter = int(MaxIter)
    m = np.zeros((T, R))
    v = np.zeros((T, R))
    xtab = np.zeros((R, T, MaxIter))
    ftab = np.full((R, MaxIter), np.nan)
    done = np.zeros(R, dtype=np.int64)
    xxp = np.linspace(-(batch - 1) / 2.0, (batch - 1) / 2.0, batch)
    it = 0
    for it in range(1, MaxIter + 1):
        F, G = fun_batch(X, it)
        act = done == 0
        ftab[act, it - 1] = np.asarray(F)[act]
        G = np.asarray(G, dtype=np.float64).reshape(T, R)
        m[:, act] = b1 * m[:, act] + (1 - b1) * G[:, act]
        v[:, act] = b2 * v[:, act] + (1 - b2) * G[:, act] ** 2
        step = ms["min"] + (ms["max"] - ms["min"]) * math.exp(-it / ms["decay"])
        X[:, act] = X[:, act] - step * (m[:, act] / (1 - b1**it)) / (np.sqrt(v[:, act] / (1 - b2**it)) + fudge)
        xtab[act, :, it - 1] = X[:, act].T
        if it % batch == 0 and it >= 2 * batch:
            for r in np.nonzero(act)[0]:
                slope, svar = _polyfit1(xxp, ftab[r, it - batch:it])
                slope_err = math.sqrt(svar + TolFun**2)
                slope_err_max = math.sqrt(svar + TolFun_max**2)
                dx = math.sqrt(np.sum((np.mean(xtab[r, :, it - batch:it], axis=1) - np.mean(xtab[r, :, it - 2 * batch:it - batch], axis=1)) ** 2 / batch))
                if (dx < TolX and abs(slope) < slope_err_max) or (abs(slope) < slope_err and dx < TolX_max):
                    done[r] = it
            if np.all(done != 0):
                break
    iters = np.where(done != 0, done, it)
    xo = np.zeros((T, R))
    fo = np.zeros(R)
    xl, fl = [], []
    for r in range(R):
        n = int(iters[r])
        nb = min(batch, n)
        xo[:, r] = np.mean(xtab[r, :, n - nb:n], axis=1)
        fo[r] = float(np.mean(ftab[r, n - nb:n]))
        xl.append(xtab[r, :, :n].copy())
        fl.append(ftab[r, :n].copy())
    return xo, fo, xl, fl, iters


def cmaes_batched(fun_batch, x0, insigma, *, TolX, TolFun, TolHistFun, MaxFunEvals=np.inf, MaxIter=None, rng=None, popsize=None):
    """A plain (mu/mu_w, lambda)-CMA-ES (Hansen's tutorial formulation: rank-one + rank-mu update, cumulative step-size
    adaptation) minimising a NOISY objective whose whole population is evaluated by ONE batched device pass per generation
    (``fun_batch(X (T x lambda)) -> F (lambda)``).  It stands where misc/vpoptimize_vbmc.m:152-153 calls the third-party
    ``cmaes_modded`` (utils/cmaes_modded.m, 3070 lines, not restated): same role, same stopping tolerances (TolX on
    sigma*sqrt(diag C) and sigma*pc, TolFun / TolHistFun on the recent best values, MaxFunEvals), default population
    4 + floor(3 ln N), but none of its restarts, active-CMA or uncertainty-handling extensions.  Separable (diagonal)
    covariance for N > 200 so that a generation stays O(N lambda)."""
    rng = np.random.default_rng(0) if rng is None else rng
    xmean = np.asarray(x0, dtype=np.float64).reshape(-1).copy()
    N = xmean.size
    insigma = np.broadcast_to(np.asarray(insigma, dtype=np.float64).reshape(-1), (N,)).copy()
    lam = int(popsize or (4 + math.floor(3 * math.log(N))))
    mu = lam // 2
    wts = math.log(mu + 0.5) - np.log(np.arange(1, mu + 1))
    wts = wts / np.sum(wts)
    mueff = 1.0 / np.sum(wts**2)
    cc = (4 + mueff / N) / (N + 4 + 2 * mueff / N)
    cs = (mueff + 2) / (N + mueff + 5)
    c1 = 2 / ((N + 1.3) ** 2 + mueff)
    cmu = min(1 - c1, 2 * (mueff - 2 + 1 / mueff) / ((N + 2) ** 2 + mueff))
    damps = 1 + 2 * max(0.0, math.sqrt((mueff - 1) / (N + 1)) - 1) + cs
    chiN = math.sqrt(N) * (1 - 1 / (4 * N) + 1 / (21 * N * N))
    diag_only = N > 200
    if diag_only:   # separable CMA-ES (Ros & Hansen 2008): learning rates scaled by (N + 2) / 3
        c1 = min(1.0, c1 * (N + 2) / 3)
        cmu = min(1 - c1, cmu * (N + 2) / 3)
    sigma = float(np.max(insigma))
    scale = insigma / sigma            # initial coordinate-wise standard deviations as a diagonal C
    Cd = scale**2                      # diagonal of C (diag_only) ...
    C = np.diag(scale**2)              # ... or the full matrix
    B = np.eye(N)
    Dv = scale.copy()
    pc = np.zeros(N)
    ps = np.zeros(N)
    MaxIter = int(MaxIter or 1e3 * (N + 5) ** 2 / math.sqrt(lam))
    hist = []
    evals, gen, eig_at = 0, 0, 0
    best_x, best_f = xmean.copy(), np.inf
    stop = "MaxIter"
    while gen < MaxIter:
        gen += 1
        Z = rng.standard_normal((N, lam))
        Y = (np.sqrt(Cd)[:, None] * Z) if diag_only else (B @ (Dv[:, None] * Z))
        X = xmean[:, None] + sigma * Y
        F = np.asarray(fun_batch(np.asfortranarray(X)), dtype=np.float64).reshape(-1)
        evals += lam
        F = np.where(np.isfinite(F), F, np.inf)
        order = np.argsort(F, kind="stable")
        if F[order[0]] < best_f:
            best_f, best_x = float(F[order[0]]), X[:, order[0]].copy()
        ysel = Y[:, order[:mu]]
        yw = ysel @ wts
        xmean = xmean + sigma * yw
        if diag_only:
            ps = (1 - cs) * ps + math.sqrt(cs * (2 - cs) * mueff) * (yw / np.sqrt(Cd))
        else:
            ps = (1 - cs) * ps + math.sqrt(cs * (2 - cs) * mueff) * (B @ ((B.T @ yw) / Dv))
        hsig = np.linalg.norm(ps) / math.sqrt(1 - (1 - cs) ** (2 * gen)) / chiN < 1.4 + 2 / (N + 1)
        pc = (1 - cc) * pc + (hsig * math.sqrt(cc * (2 - cc) * mueff)) * yw
        dh = (1 - hsig) * cc * (2 - cc)
        if diag_only:
            Cd = (1 - c1 - cmu) * Cd + c1 * (pc**2 + dh * Cd) + cmu * ((ysel**2) @ wts)
        else:
            C = (1 - c1 - cmu) * C + c1 * (np.outer(pc, pc) + dh * C) + cmu * ((ysel * wts[None, :]) @ ysel.T)
            if evals - eig_at > lam / (c1 + cmu) / N / 10:
                eig_at = evals
                C = np.triu(C) + np.triu(C, 1).T
                ev, B = np.linalg.eigh(C)
                Dv = np.sqrt(np.maximum(ev, 1e-300))
        sigma = sigma * math.exp((cs / damps) * (np.linalg.norm(ps) / chiN - 1))
        hist.append(float(F[order[0]]))
        sd = sigma * (np.sqrt(Cd) if diag_only else np.sqrt(np.maximum(np.diag(C), 0.0)))
        if evals >= MaxFunEvals:
            stop = "MaxFunEvals"
            break
        if np.all(sd < TolX) and np.all(sigma * np.abs(pc) < TolX):
            stop = "TolX"
            break
        nh = 10 + int(math.ceil(30 * N / lam))
        if gen > 2 and max(hist[-1], float(F[order[-1]])) - min(hist[-1], float(F[order[0]])) < TolFun and \
                (max(hist[-min(nh, len(hist)):]) - min(hist[-min(nh, len(hist)):])) < TolFun:
            stop = "TolFun"
            break
        if len(hist) > nh and (max(hist[-nh:]) - min(hist[-nh:])) < TolHistFun:
            stop = "TolHistFun"
            break
    return xmean, {"stop": stop, "generations": gen, "evals": evals, "best_x": best_x, "best_f": best_f, "sigma": sigma}


def vpoptimize_vbmc(Nfastopts, Nslowopts, vp, gp, K=None, optimState=None, options=None, prnt=0, *, rng=None, seed=0,
                    engine=None, shard=None, device_adam=True, trace=None):
    """[vp,varss,pruned] = vpoptimize_vbmc(Nfastopts,Nslowopts,vp,gp,K,optimState,options,prnt)
    (misc/vpoptimize_vbmc.m:1-254).

    Execution shape: the Nslowopts chains are selected as the reference does (:53-61) and then advanced TOGETHER -- one
    vbmc_adam_batch (the whole Adam loop on the device) or, with device_adam=False, fminadam_lockstep with one batched
    objective call per iteration; their 2*Nslowopts eval_fullelcbo calls (midpoint :134, endpoint :165) are ONE batched
    pass; the pruning loop (:190-243) follows the reference step by step (its evaluations depend on each other).

    Branches: NSentK > 0 with ELCBOWeight = 0 -> Adam (:108-135).  ELCBOWeight ~= 0 -> the reference has no gradient of the
    full variance and switches to CMA-ES on the value (:38-46,137-160): mirrored with cmaes_batched, each generation one
    batched value-only pass with compute_var = 1.  NSentK = 0 (deterministic entropy: EntropySwitch or K = 1) -> the reference
    calls fminunc from MATLAB's Optimization Toolbox (:73-81); SciPy's BFGS drives the same device objective here, with the
    device gradient or (ELCBOWeight ~= 0) forward differences whose T + 1 points are one batched pass.

    Device-stream schedule (``seed`` = s; tests replay it through ``trace``): sieve s; Adam iteration it of chain group g
    (s << 20) + (1 << 16) + (g << 12) + it; eval_fullelcbo batch (s << 20) + (2 << 16); pruning evaluation number c
    (s << 20) + (3 << 16) + c; CMA-ES generation n of chain iOpt (s << 20) + (4 << 16) + (iOpt << 12) + n."""
    options = dict(DEFAULT_OPTIONS, **(options or {}))
    optimState = dict(optimState or {})
    rng = np.random.default_rng(0) if rng is None else rng
    K = vp["K"] if K is None else K
    optimState.setdefault("Warmup", not vp["optimize_weights"])
    optimState.setdefault("temperature", 1)
    vp0_vec, vp0_type, elcbo_beta, compute_var, NSentK, _, _ = vpsieve_vbmc(Nfastopts, Nslowopts, vp, gp, optimState, options, K,
                                                                            rng=rng, seed=seed, engine=engine, shard=shard)
    vp, thetabnd = vpbounds(vp, gp, options, K)
    gradient_available = not compute_var          # :38
    optimizer = str(options["StochasticOptimizer"]).lower() if gradient_available else "cmaes"   # :44
    if NSentK == 0:
        # deterministic entropy (EntropySwitch or K = 1, :73-106): the reference calls fminunc (MATLAB's Optimization Toolbox,
        # quasi-Newton BFGS, TolFun = DetEntTolOpt, MaxFunEvals = 50 (D + 2)); here SciPy's BFGS stands in its place, one
        # device evaluation (value + gradient, entlb_vbmc entropy) per objective call
        # With ELCBOWeight ~= 0 (no gradient of the full variance, :38-46) fminunc differences the value (GradObj off, forward
        # differences): the T + 1 points of one difference gradient are ONE batched value-only device pass here.
        optimizer = "bfgs"
    if optimizer not in ("adam", "cmaes", "bfgs"):
        raise ValueError("vbmc:VPoptimize Unknown stochastic optimizer.")
    D = vp["D"]
    vp0_type = list(vp0_type)
    vp0_vec = list(vp0_vec)
    sbase = int(seed) << 20

    # ---- starting points (:50-69)
    starts, theta0s = [], []
    for iOpt in range(1, Nslowopts + 1):
        if Nslowopts == 1:
            idx = 0
        elif Nslowopts == 2:
            idx = next(i for i, t in enumerate(vp0_type) if (t == 1 if iOpt == 1 else t in (2, 3)))
        else:
            idx = next(i for i, t in enumerate(vp0_type) if t == ((iOpt - 1) % 3) + 1)
        vp0 = rescale_params(vp0_vec[idx])
        del vp0_type[idx], vp0_vec[idx]
        starts.append(vp0)
        theta0s.append(get_vptheta(vp0)[0])

    # ---- optimisation
    thetaopt = [None] * Nslowopts
    theta_mid = [None] * Nslowopts
    if optimizer == "bfgs":
        from scipy.optimize import minimize

        for i, vp0 in enumerate(starts):
            nev = [0]

            def fun(th, vp0=vp0):
                th = np.asarray(th, dtype=np.float64)
                if gradient_available:
                    nev[0] += 1
                    r = negelcbo_batch(th, elcbo_beta, vp0, gp, 0, True, 0, thetabnd, engine=engine, outputs=("F", "dF"))
                    return float(r["F"][0]), r["dF"][:, 0].copy()
                # fminunc's forward differences: step sqrt(eps) * max(|x|, TypicalX = 1), signed like x
                h = np.sqrt(np.finfo(np.float64).eps) * np.where(th < 0, -1.0, 1.0) * np.maximum(np.abs(th), 1.0)
                X = np.repeat(th[:, None], th.size + 1, axis=1)
                X[np.arange(th.size), np.arange(1, th.size + 1)] += h
                h = X[np.arange(th.size), np.arange(1, th.size + 1)] - th        # the step actually taken in floating point
                nev[0] += th.size + 1
                F = negelcbo_batch(np.asfortranarray(X), elcbo_beta, vp0, gp, 0, False, int(bool(compute_var)), thetabnd, engine=engine,
                                   outputs=("F",))["F"]
                return float(F[0]), (F[1:] - F[0]) / h

            # MaxFunEvals = 50 (D + 2) counts every differenced point in fminunc (:76): the iteration cap follows from it
            maxit = 50 * (D + 2) if gradient_available else max(2, (50 * (D + 2)) // (theta0s[i].size + 1))
            res = minimize(fun, theta0s[i], jac=True, method="BFGS", options={"gtol": options["DetEntTolOpt"], "maxiter": maxit})
            thetaopt[i] = np.asarray(res.x, dtype=np.float64)
            if trace is not None:
                trace.append({"kind": "bfgs", "slot": i, "nfev": int(nev[0]), "fun": float(res.fun), "fd": not gradient_available})
    elif optimizer == "adam":
        ms = {"min": min(options["SGDStepSize"], 0.001)}
        scaling = min(0.1, options["SGDStepSize"] * 10) if (optimState["Warmup"] or not vp["optimize_weights"]) else min(0.1, options["SGDStepSize"])
        ms["max"] = max(ms["min"], scaling)
        ms["decay"] = 200
        MaxIter = min(options["MaxIterStochastic"] or 100 * (2 + D), 10000)
        for gi, members in enumerate(_group_by_fixed(starts)):
            X0 = np.asfortranarray(np.stack([theta0s[i] for i in members], axis=1))
            sd = sbase + (1 << 16) + (gi << 12)
            vpg = starts[members[0]]
            xmid = None
            if device_adam:  # the whole Adam loop on the device (vbmc_adam_batch), no host round trip per evaluation; the best midpoint
                # of each chain (:133) is picked inside the library, the iterate tables (T x MaxIter x R) stay on the device
                xo, _, xmid, _, its = fminadam_device(X0, elcbo_beta, vpg, gp, NSentK, thetabnd, options["TolFunStochastic"], MaxIter, ms,
                                                      seed=sd, engine=engine, tables=False)
            else:
                def fun_batch(X, it, vpg=vpg, sd=sd):
                    r = negelcbo_batch(X, elcbo_beta, vpg, gp, NSentK, True, 0, thetabnd, seed=sd + it, engine=engine, outputs=("F", "dF"))
                    return r["F"], r["dF"]

                xo, _, xt, ft, its = fminadam_lockstep(fun_batch, X0, options["TolFunStochastic"], MaxIter, ms)
            for r, i in enumerate(members):
                thetaopt[i] = xo[:, r].copy()
                if options["ELCBOmidpoint"]:
                    theta_mid[i] = (xmid[:, r] if xmid is not None else xt[r][:, int(np.argmin(ft[r]))]).copy()   # :133 [~,idx_mid] = min(fval_lst)
                if trace is not None:
                    trace.append({"kind": "adam", "slot": i, "seed": sd, "r": r, "R": len(members), "K": K, "Ns": NSentK, "iters": int(its[r])})
    else:
        b = vp["bounds"]
        ins = []
        if vp["optimize_mu"]:
            ins.append(np.tile(b["mu_ub"] - b["mu_lb"], K))
        if vp["optimize_sigma"]:
            ins.append(np.ones(K))
        if vp["optimize_lambda"]:
            ins.append(np.ones(D))
        if vp["optimize_weights"]:
            ins.append(np.ones(K))
        insigma = np.concatenate(ins)
        cv = int(bool(compute_var))
        for i, vp0 in enumerate(starts):
            gen = [0]

            def fun_batch(X, vp0=vp0, i=i):
                gen[0] += 1
                r = negelcbo_batch(X, elcbo_beta, vp0, gp, NSentK, False, cv, thetabnd, seed=sbase + (4 << 16) + ((i + 1) << 12) + gen[0],
                                   engine=engine, outputs=("F",))
                return r["F"]

            thetaopt[i], info = cmaes_batched(fun_batch, theta0s[i], insigma, TolX=1e-6 * float(np.max(insigma)), TolFun=1e-4, TolHistFun=1e-5,
                                              MaxFunEvals=options.get("CMAESMaxFunEvals") or 1000 * (D + 2), rng=rng)   # :145-149
            # (the reference leaves MaxFunEvals at Inf and relies on cmaes_modded's uncertainty handling, Noise.on = 1, to end a
            # run on the noisy objective; the plain CMA-ES here has none, hence a cap: five times the 200 (D + 2) the reference
            # allows its own deterministic CMA-ES fallback, :97)
            if trace is not None:
                trace.append(dict(kind="cmaes", slot=i, **{k: info[k] for k in ("stop", "generations", "evals")}))

    # ---- full ELCBO at the midpoints and endpoints, one batched pass (:131-135,165)
    slots, th_list, vp_list = [], [], []
    for i in range(Nslowopts):
        if theta_mid[i] is not None:
            slots.append(2 * i)
            th_list.append(theta_mid[i])
            vp_list.append(starts[i])
        slots.append(2 * i + 1)
        th_list.append(thetaopt[i])
        vp_list.append(starts[i])
    st = eval_fullelcbo_batch(th_list, vp_list, gp, elcbo_beta, options, seed=sbase + (2 << 16), engine=engine, trace=trace, labels=slots)
    nelcbo = np.full(2 * Nslowopts, np.inf)      # elbostats of :33 -- empty slots stay Inf
    by_slot = {}
    for sl, stat, v0 in zip(slots, st, vp_list):
        nelcbo[sl] = stat["nelcbo"]
        by_slot[sl] = (stat, v0)

    # ---- best ELCBO (:174-192)
    best = int(np.argmin(nelcbo))              # first minimum, like MATLAB's min
    s, vpbest = by_slot[best]
    elbo, elbo_sd = -s["nelbo"], math.sqrt(s["varF"])
    G, H, varss, varG, varH = s["G"], s["H"], s["varss"], s["varG"], s["varH"]
    I_sk = s["I_sk"].copy()
    J_sjk = s["J_sjk"].copy() if s["J_sjk"] is not None else None
    vp = rescale_params(vpbest, s["theta"])
    vp["temperature"] = optimState["temperature"]

    # ---- pruning of mixture components (:196-243)
    pruned = 0
    if vp["optimize_weights"]:
        alreadychecked = np.zeros(vp["K"], dtype=bool)
        count = 0
        # square copies of the per-component terms for the recombination below (the reported J_sjk loses only its third dimension on
        # a removal, as the reference's does, :233)
        Ic = I_sk.copy()
        Jc = J_sjk.copy() if (J_sjk is not None and not options.get("SkipELBOVariance")) else None
        while np.any((vp["w"] < options["TolWeight"]) & ~alreadychecked):
            cand = np.nonzero((vp["w"] < options["TolWeight"]) & ~alreadychecked)[0]
            idx = int(cand[int(rng.integers(cand.size))])          # idx(randi(numel(idx)))
            vpp = copy_vp(vp)
            vpp["w"] = np.delete(vpp["w"], idx)
            if "eta" in vpp:
                vpp["eta"] = np.delete(vpp["eta"], idx)
            vpp["sigma"] = np.delete(vpp["sigma"], idx)
            vpp["mu"] = np.delete(vpp["mu"], idx, axis=1)
            vpp["K"] = vpp["K"] - 1
            theta_p, vpp = get_vptheta(vpp)
            count += 1
            if Jc is not None and not options.get("VBMCHipPruneFullEval"):
                # The expected log joint of the candidate and its variance are functions of the per-component terms the last full
                # evaluation already returned -- I_sk and J_sjk of the components that stay, with the candidate's weights: the reference's
                # own misc/gplogjoint_weights.m recombines them the same way -- so only the Monte-Carlo entropy needs the device
                # (same stream as the full evaluation would draw: same seed, same kernel).  0.28 -> 0.08 ms per attempt (round 3).
                keepk = np.delete(np.arange(Ic.shape[1]), idx)
                sp = _stats_from_cache(Ic[:, keepk], Jc[:, keepk][:, :, keepk], np.asarray(vpp["w"], dtype=np.float64).reshape(-1))
                NSentFineK = int(math.ceil(evaloption(options["NSentFine"], vpp["K"]) / vpp["K"]))
                sd = sbase + (3 << 16) + count
                Hn = float(negelcbo_batch(np.asarray(theta_p, dtype=np.float64).reshape(-1, 1), 0, vpp, None, NSentFineK, False, 0, None,
                                          seed=sd, engine=engine, outputs=("H",))["H"][0])
                sp.update(H=Hn, nelbo=-sp["G"] - Hn, varH=0.0, theta=np.asarray(theta_p, dtype=np.float64).reshape(-1).copy())
                sp["nelcbo"] = sp["nelbo"] + elcbo_beta * math.sqrt(sp["varF"])
                if trace is not None:
                    trace.append({"kind": "prune", "slot": count, "seed": sd, "r": 0, "R": 1, "K": vpp["K"], "Ns": NSentFineK})
            else:
                sp = eval_fullelcbo_batch([theta_p], [vpp], gp, elcbo_beta, options, seed=sbase + (3 << 16) + count, engine=engine,
                                          trace=trace, kind="prune", labels=[count])[0]
            elbo_p, elbo_p_sd = -sp["nelbo"], math.sqrt(sp["varF"])
            delta_elcbo = abs((elbo_p - options["ELCBOImproWeight"] * elbo_p_sd) - (elbo - options["ELCBOImproWeight"] * elbo_sd))
            thr = options["TolImprovement"] * evaloption(options["PruningThresholdMultiplier"], K)
            if delta_elcbo < thr:
                vp = vpp
                elbo, elbo_sd = elbo_p, elbo_p_sd
                G, H, varss, varG, varH = sp["G"], sp["H"], sp["varss"], sp["varG"], sp["varH"]
                pruned += 1
                if Jc is not None:
                    keepk = np.delete(np.arange(Ic.shape[1]), idx)
                    Ic, Jc = Ic[:, keepk], Jc[:, keepk][:, :, keepk]
                alreadychecked = np.delete(alreadychecked, idx)
                I_sk = np.delete(I_sk, idx, axis=1)                 # I_sk(:,idx) = []
                if J_sjk is not None:
                    J_sjk = np.delete(J_sjk, idx, axis=2)           # J_sjk(:,:,idx) = []  (third dimension only, as the reference)
            else:
                alreadychecked[idx] = True

    vp["stats"] = {"elbo": elbo, "elbo_sd": elbo_sd, "elogjoint": G, "elogjoint_sd": math.sqrt(varG), "entropy": H,
                   "entropy_sd": math.sqrt(varH), "stable": False, "I_sk": I_sk, "J_sjk": J_sjk}
    return vp, varss, pruned
