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

import copy
import math

import numpy as np

from .elbo import fminadam_device, negelcbo_batch
from .vp import DEFAULT_OPTIONS, evaloption, get_vptheta, rescale_params, vpbounds

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
        v = copy.deepcopy(vp)
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


def _template_vp(vp0):
    """All candidates share flags/shapes; non-optimised groups are read from this vp on the device."""
    return vp0


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
    vp = copy.deepcopy(vp)
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


def eval_fullelcbo(theta, vp, gp, beta, options, *, seed=0, engine=None):
    """eval_fullelcbo (misc/vpoptimize_vbmc.m:257-305): NSentFine samples, full variance, per-component terms."""
    K = vp["K"]
    NSentFineK = int(math.ceil(evaloption(options["NSentFine"], K) / K))
    cv = 0 if options.get("SkipELBOVariance") else 1
    out = negelcbo_batch(np.asarray(theta).reshape(-1), 0, vp, gp, NSentFineK, False, cv, None, separate_K=True, seed=seed, engine=engine)
    varF = float(out["varG"][0]) if cv else 0.0
    return {"nelbo": float(out["F"][0]), "G": float(out["G"][0]), "H": float(out["H"][0]), "varF": varF, "varG": varF,
            "varH": 0.0, "varss": float(out["varGss"][0]) if cv else 0.0, "nelcbo": float(out["F"][0]) + beta * math.sqrt(varF),
            "theta": np.asarray(theta).reshape(-1).copy(), "I_sk": out["I_sk"][:, :, 0].copy(),
            "J_sjk": out["J_sjk"][:, :, :, 0].copy() if cv else None}


def vpoptimize_vbmc(Nfastopts, Nslowopts, vp, gp, K=None, optimState=None, options=None, prnt=0, *, rng=None, seed=0,
                    engine=None, shard=None, device_adam=True):
    """[vp,varss,pruned] = vpoptimize_vbmc(Nfastopts,Nslowopts,vp,gp,K,optimState,options,prnt)
    (misc/vpoptimize_vbmc.m:1-254), stochastic (Adam) path with NSentK > 0; the deterministic-entropy
    fminunc branch (:73-106) needs MATLAB's Optimization Toolbox and is not mirrored."""
    options = dict(DEFAULT_OPTIONS, **(options or {}))
    optimState = dict(optimState or {})
    rng = np.random.default_rng(0) if rng is None else rng
    K = vp["K"] if K is None else K
    optimState.setdefault("Warmup", not vp["optimize_weights"])
    optimState.setdefault("temperature", 1)
    vp0_vec, vp0_type, elcbo_beta, compute_var, NSentK, _, _ = vpsieve_vbmc(Nfastopts, Nslowopts, vp, gp, optimState, options, K,
                                                                            rng=rng, seed=seed, engine=engine, shard=shard)
    vp, thetabnd = vpbounds(vp, gp, options, K)
    if compute_var or NSentK == 0:
        raise NotImplementedError("only the gradient-available stochastic path (ELCBOWeight = 0, NSentK > 0) is mirrored")
    D = vp["D"]
    vp0_type = list(vp0_type)
    stats = []
    for iOpt in range(1, Nslowopts + 1):
        if Nslowopts == 1:
            idx = 0
        elif Nslowopts == 2:
            idx = next(i for i, t in enumerate(vp0_type) if (t == 1 if iOpt == 1 else t in (2, 3)))
        else:
            idx = next(i for i, t in enumerate(vp0_type) if t == ((iOpt - 1) % 3) + 1)
        vp0 = rescale_params(vp0_vec[idx])
        del vp0_type[idx], vp0_vec[idx]
        theta0, _ = get_vptheta(vp0)
        ms = {"min": min(options["SGDStepSize"], 0.001)}
        scaling = min(0.1, options["SGDStepSize"] * 10) if (optimState["Warmup"] or not vp["optimize_weights"]) else min(0.1, options["SGDStepSize"])
        ms["max"] = max(ms["min"], scaling)
        ms["decay"] = 200
        MaxIter = min(options["MaxIterStochastic"] or 100 * (2 + D), 10000)
        counter = [0]

        def fun(th, vp0=vp0):
            counter[0] += 1
            r = negelcbo_batch(th, elcbo_beta, vp0, gp, NSentK, True, 0, thetabnd, seed=(seed << 20) + (iOpt << 16) + counter[0], engine=engine)
            return float(r["F"][0]), r["dF"][:, 0]

        if device_adam:  # the whole Adam loop on the device (vbmc_adam_batch), no host round trip per evaluation
            xo, _, xt, ft, _ = fminadam_device(theta0, elcbo_beta, vp0, gp, NSentK, thetabnd, options["TolFunStochastic"], MaxIter, ms,
                                               seed=(seed << 20) + (iOpt << 16), engine=engine)
            thetaopt, theta_lst, fval_lst = xo[:, 0], xt[0], ft[0]
        else:
            thetaopt, _, theta_lst, fval_lst, _ = fminadam(fun, theta0, None, None, options["TolFunStochastic"], MaxIter, ms)
        if options["ELCBOmidpoint"]:
            imid = int(np.argmin(fval_lst))
            st = eval_fullelcbo(theta_lst[:, imid], vp0, gp, elcbo_beta, options, seed=seed + 7 * iOpt, engine=engine)
            st["vp0"] = vp0
            stats.append(st)
        st = eval_fullelcbo(thetaopt, vp0, gp, elcbo_beta, options, seed=seed + 7 * iOpt + 1, engine=engine)
        st["vp0"] = vp0
        stats.append(st)
    best = int(np.argmin([s["nelcbo"] for s in stats]))
    s = stats[best]
    vp = rescale_params(s["vp0"], s["theta"])
    vp["temperature"] = optimState["temperature"]
    vp["stats"] = {"elbo": -s["nelbo"], "elbo_sd": math.sqrt(s["varF"]), "elogjoint": s["G"], "elogjoint_sd": math.sqrt(s["varG"]),
                   "entropy": s["H"], "entropy_sd": 0.0, "stable": False, "I_sk": s["I_sk"], "J_sjk": s["J_sjk"]}
    return vp, s["varss"], 0
