"""gplite GP surrogate on the GPU behind the reference's call surface.

``gplite_post(hyp, X, y, covfun, meanfun, noisefun, s2)`` (gplite/gplite_post.m:1),
``gplite_pred(gp, Xstar, ystar, s2star, ssflag)`` (gplite/gplite_pred.m:1) and
``gplite_nlZ(hyp, gp, hprior)`` (gplite/gplite_nlZ.m:1), ``gplite_hypprior(hyp, hprior)``
(gplite/gplite_hypprior.m:1) and ``sq_dist(a, b)`` (utils/sq_dist.m:14) keep the reference's positional arguments.  ``gp`` is a
dict with the reference's field names; its ``post`` list holds the per-hyper-sample
{hyp, alpha, sW, L, sn2_mult, Lchol} exactly like ``gp.post(s)``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import DeviceGP, f64, ptr
from .elbo import default_engine


def _nnoise(noisefun):
    return int(noisefun[0] == 1) + int(noisefun[1] == 2) + 2 * int(len(noisefun) > 2 and noisefun[2] == 1)


def _nmean(meanfun, D):
    return {0: 0, 1: 1, 4: 2 * D + 1}.get(int(meanfun), -1)


def sq_dist(a, b=None, *, engine=None):
    """C = sq_dist(a, b): pairwise squared distances between the columns of a (D x n) and b (D x m)."""
    engine = engine or default_engine()
    ctx = engine.ctx
    a = f64(a)
    D, n = a.shape
    if b is None:
        m, bp = n, None
    else:
        b = f64(b)
        if b.shape[0] != D:
            raise ValueError("Error: column lengths must agree.")  # sq_dist.m:35
        m, bp = b.shape[1], ptr(b)
    Cm = np.zeros((n, m), dtype=np.float64, order="F")
    ctx.check(ctx.lib.vbmc_sq_dist(ctx.h, D, n, m, ptr(a), bp, ptr(Cm)))
    return Cm


def gplite_post(hyp, X, y, covfun=1, meanfun=1, noisefun=None, s2=None, *, need_L=True, engine=None):
    """gp = gplite_post(hyp,X,y,covfun,meanfun,noisefun,s2): full posterior for every hyper-sample.

    ``need_L=False`` keeps the N x N x S factors on the device only (``post[s]["L"]`` is None): every accelerated consumer
    (gplite_pred, the ELBO, the acquisition sweep, the rank-one append) reads the device copy, and the 8 N^2 S bytes of
    readback -- 25.6 MB at N = 400, S = 20, most of the call's wall time -- are skipped.

    Only the VBMC configuration is accelerated: covfun 1 (SE-ARD), meanfun in {0, 1, 4}; anything
    else raises VbmcUnsupported so a caller can fall through to the reference implementation.
    """
    engine = engine or default_engine()
    ctx = engine.ctx
    X = f64(X)
    N, D = X.shape
    y = f64(np.asarray(y, dtype=np.float64).reshape(-1))
    hyp = f64(np.asarray(hyp, dtype=np.float64))
    if hyp.ndim == 1:
        hyp = f64(hyp.reshape(-1, 1))
    Nhyp, S = hyp.shape
    if covfun is None:
        covfun = 1
    if np.ndim(covfun) and len(covfun):
        covfun = covfun[0]
    if int(covfun) != 1:
        from ._lib import VBMC_ERR_UNSUPPORTED, VbmcUnsupported

        raise VbmcUnsupported(VBMC_ERR_UNSUPPORTED, "only the SE-ARD covariance (covfun 1) is accelerated")
    if meanfun is None:
        meanfun = 1  # gplite_post.m:103
    if noisefun is None:
        noisefun = (1, 0, 0) if s2 is None else (1, 1, 0)  # :104-106
    noisefun = tuple(int(v) for v in noisefun) + (0,) * (3 - len(noisefun))
    s2a = None if s2 is None else f64(np.asarray(s2, dtype=np.float64).reshape(-1))
    alpha = np.empty((N, S), order="F")      # all three are overwritten in full by the library
    L = np.empty((N, N, S), order="F") if need_L else None
    sW = np.empty((N, S), order="F")
    mult = np.zeros(S)
    lch = np.zeros(S, dtype=np.uint8)
    nf = (C.c_int32 * 3)(*noisefun)
    h = C.c_void_p()
    ctx.check(ctx.lib.vbmc_gp_post(ctx.h, N, D, S, Nhyp, int(meanfun), nf, ptr(X), ptr(y), ptr(s2a), ptr(hyp), ptr(alpha),
                                   ptr(L), ptr(sW), ptr(mult), lch.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(h)))
    gp = {
        "X": X, "y": y, "s2": s2a, "covfun": 1, "meanfun": int(meanfun), "noisefun": noisefun,
        "Ncov": D + 1, "Nnoise": _nnoise(noisefun), "Nmean": _nmean(meanfun, D), "meanfun_extras": None, "intmeanfun": 0,
        # L[:, :, s] is a contiguous (column-major) view of the N x N x S block written by the library: no second copy
        "post": [{"hyp": hyp[:, s].copy(), "alpha": alpha[:, s].copy(), "sW": sW[:, s].copy(), "L": L[:, :, s] if need_L else None,
                  "sn2_mult": float(mult[s]), "Lchol": bool(lch[s])} for s in range(S)],
    }
    dgp = DeviceGP.from_handle(ctx, h, N, D, S)
    engine._remember(gp, dgp, True)
    return gp


def _device_gp_with_noise(engine, gp):
    dgp = engine.device_gp(gp, need_L=True)
    dgp.set_noise(gp["noisefun"], [p["sn2_mult"] for p in gp["post"]])
    return dgp


def gplite_pred(gp, Xstar, ystar=None, s2star=None, ssflag=False, nowarpflag=False, nargout=4, *, engine=None):
    """[ymu,ys2,fmu,fs2,lp] = gplite_pred(gp,Xstar,ystar,s2star,ssflag).  ``nargout=5`` with ``ystar`` adds the log
    predictive density lp (Nstar x S, per hyper-sample also when the other outputs are averaged: gplite_pred.m:124-127)."""
    if nargout > 4:
        Ns = np.asarray(Xstar).shape[0]
        if ystar is not None and np.size(ystar) and np.asarray(ystar).reshape(-1).shape[0] != Ns:
            raise ValueError("gplite_pred:ydimmismatch YSTAR should be empty or a column vector of NSTAR observations.")
        ymu_s, ys2_s = gplite_pred(gp, Xstar, ystar, s2star, True, nowarpflag, 2, engine=engine)
        lp = None
        if ystar is not None and np.size(ystar):
            ymu_s = np.asarray(ymu_s).reshape(Ns, -1)
            ys2_s = np.asarray(ys2_s).reshape(Ns, -1)
            yv = np.asarray(ystar, dtype=np.float64).reshape(-1, 1)
            lp = -0.5 * (yv - ymu_s) ** 2 / ys2_s - 0.5 * np.log(2 * np.pi * ys2_s)   # O(Nstar S) on the host
        return tuple(gplite_pred(gp, Xstar, ystar, s2star, ssflag, nowarpflag, 4, engine=engine)) + (lp,)
    engine = engine or default_engine()
    ctx = engine.ctx
    Xs = f64(Xstar)
    Nstar = Xs.shape[0]
    if s2star is not None and np.size(s2star) and np.asarray(s2star).reshape(-1).shape[0] != Nstar:
        raise ValueError("gplite_pred:s2dimmismatch S2STAR should be empty or a column vector of NSTAR estimated variances.")
    s2s = None if s2star is None or np.size(s2star) == 0 else f64(np.asarray(s2star, dtype=np.float64).reshape(-1))
    if ystar is not None and np.size(ystar) and np.asarray(ystar).reshape(-1).shape[0] != Nstar:
        raise ValueError("gplite_pred:ydimmismatch YSTAR should be empty or a column vector of NSTAR observations.")
    # ystar only matters for output-dependent noise at the test points (gplite_noisefun.m:198-207)
    ys = None if ystar is None or np.size(ystar) == 0 else f64(np.asarray(ystar, dtype=np.float64).reshape(-1))
    dgp = _device_gp_with_noise(engine, gp)
    S = dgp.S
    per = bool(ssflag) or S == 1
    shape = (Nstar, S) if (per and S > 1) else (Nstar,)
    outs = [np.zeros((Nstar, S) if per else (Nstar,), order="F") for _ in range(4)]
    ctx.check(ctx.lib.vbmc_gp_pred(ctx.h, dgp.h, Nstar, ptr(Xs), ptr(ys), ptr(s2s), 1 if per else 0, ptr(outs[0]), ptr(outs[1]),
                                   ptr(outs[2]), ptr(outs[3])))
    outs = [o.reshape(shape, order="F") if per else o for o in outs]
    return tuple(outs[: max(1, nargout)])


def gplite_post_rank1(gp, xstar, ystar, s2star=None, *, need_L=True, engine=None):
    """gp = gplite_post(gp, xstar, ystar, [], [], [], [], 1): rank-1 append of one observation
    (gplite/gplite_post.m:173-251).  Falls back to the full update when ``s2`` is present, as the
    reference does (:76-79).  need_L=False leaves post[s]["L"] = None on the host: the updated factors stay on the
    device, where this package's own consumers (gplite_pred, acqwrapper_vbmc, negelcbo_vbmc, the next append) read them."""
    import math

    engine = engine or default_engine()
    ctx = engine.ctx
    xstar = np.asarray(xstar, dtype=np.float64).reshape(1, -1)
    ystar = float(np.asarray(ystar).reshape(-1)[0])
    has_s2 = gp.get("s2") is not None and np.size(gp["s2"]) > 0
    new_s2 = s2star is not None and np.size(s2star) > 0
    if has_s2 != new_s2:
        raise ValueError("gplite_post: the new observation %s an estimated variance s2star but gp.s2 is %s"
                         % ("has" if new_s2 else "lacks", "empty" if not has_s2 else "set"))
    if new_s2:  # heteroskedastic noise: the reference leaves the rank-1 path when the new s2 is non-empty (:76-79,86-90)
        hyp = np.stack([p["hyp"] for p in gp["post"]], axis=1)
        s2new = np.concatenate([np.asarray(gp["s2"], dtype=np.float64).reshape(-1), [float(np.asarray(s2star).reshape(-1)[0])]])
        return gplite_post(hyp, np.vstack([gp["X"], xstar]), np.concatenate([gp["y"], [ystar]]), 1, gp["meanfun"],
                           gp["noisefun"], s2new, engine=engine)
    N, D = np.asarray(gp["X"]).shape
    S = len(gp["post"])
    # [mstar, vstar] of :189 are formed inside the library from the solves of the append itself
    dgp = _device_gp_with_noise(engine, gp)
    Ncov = gp["Ncov"]
    sn2_eff = np.zeros(S)
    for s, post in enumerate(gp["post"]):
        hyp = post["hyp"]
        sn2 = math.exp(2.0 * hyp[Ncov]) if gp["noisefun"][0] == 1 else float(np.finfo(np.float64).eps)
        if len(gp["noisefun"]) > 2 and gp["noisefun"][2] == 1:
            off = Ncov + (1 if gp["noisefun"][0] == 1 else 0) + (1 if gp["noisefun"][1] == 2 else 0)
            sn2 += math.exp(2.0 * hyp[off + 1]) * max(0.0, hyp[off] - ystar) ** 2
        sn2_eff[s] = sn2 * post["sn2_mult"]                                           # :207
    # the append itself runs on the device (k_rank1_assemble) and yields a new surrogate handle; L crosses PCIe only if wanted
    Xn = f64(np.vstack([gp["X"], xstar]))
    alpha = np.empty((N + 1, S), order="F")
    Lh = np.empty((N + 1, N + 1, S), order="F") if need_L else None
    h = C.c_void_p()
    ctx.check(ctx.lib.vbmc_gp_rank1_update(ctx.h, dgp.h, ptr(Xn), C.c_double(ystar), None, None, ptr(f64(sn2_eff)), ptr(alpha),
                                           ptr(Lh) if need_L else None, C.byref(h)))
    out = {k: v for k, v in gp.items() if k != "post"}
    out["X"] = Xn
    out["y"] = np.concatenate([gp["y"], [ystar]])
    out["post"] = [{"hyp": p["hyp"].copy(), "alpha": alpha[:, s].copy(),
                    "sW": np.concatenate([p["sW"], [1.0 / math.sqrt(sn2_eff[s])]]),   # :239
                    "L": Lh[:, :, s] if need_L else None, "sn2_mult": p["sn2_mult"], "Lchol": p["Lchol"]}
                   for s, p in enumerate(gp["post"])]
    ndgp = DeviceGP.from_handle(ctx, h, N + 1, D, S)
    ndgp.set_noise(gp["noisefun"], [p["sn2_mult"] for p in gp["post"]])
    engine._remember(out, ndgp, True)
    return out


def gplite_hypprior(hyp, hprior, nargout=2):
    """[lp,dlp] = gplite_hypprior(hyp,hprior)  (gplite/gplite_hypprior.m:17-65): independent flat / Gaussian /
    Student-t log-priors per hyper-parameter.  O(Nhyp) host arithmetic, no device work."""
    from math import lgamma, pi

    hyp = np.asarray(hyp, dtype=np.float64)
    if hyp.ndim == 2 and hyp.shape[1] > 1:
        raise ValueError("gplite_hypprior:nosampling Hyperparameter log priors are available only for one-sample hyperparameter inputs.")
    hyp = hyp.reshape(-1)
    n = hyp.size
    mu = np.asarray(hprior["mu"], dtype=np.float64).reshape(-1)
    sigma = np.abs(np.asarray(hprior["sigma"], dtype=np.float64).reshape(-1))
    df = hprior.get("df")
    df = np.full(n, 7.0) if df is None or np.size(df) == 0 else np.asarray(df, dtype=np.float64).reshape(-1)
    flat = ~np.isfinite(mu) | ~np.isfinite(sigma)
    gauss = ~flat & ((df == 0) | ~np.isfinite(df)) & np.isfinite(sigma)
    stud = ~flat & (df > 0) & np.isfinite(df)
    dev = np.zeros(n)
    sel = gauss | stud
    dev[sel] = (hyp[sel] - mu[sel]) / sigma[sel]
    z2 = dev * dev
    lp = -0.5 * float(np.sum(np.log(2 * pi * sigma[gauss] ** 2) + z2[gauss]))
    dlp = np.zeros(n)
    dlp[gauss] = -dev[gauss] / sigma[gauss]
    if np.any(stud):
        nu = df[stud]
        const = np.array([lgamma(0.5 * (v + 1)) - lgamma(0.5 * v) for v in nu]) - 0.5 * np.log(pi * nu) - np.log(sigma[stud])
        lp += float(np.sum(const - 0.5 * (nu + 1) * np.log1p(z2[stud] / nu)))
        dlp[stud] = -(nu + 1) / nu / (1 + z2[stud] / nu) * dev[stud] / sigma[stud]
    return (lp, dlp) if nargout > 1 else lp


def gplite_nlZ(hyp, gp, hprior=None, nargout=2, *, engine=None):
    """[nlZ,dnlZ] = gplite_nlZ(hyp,gp,hprior)  (gplite/gplite_nlZ.m:1-72).

    ``hyp`` Nhyp (or Nhyp x 1) follows the reference: scalar nlZ and an Nhyp gradient.  Beyond the reference,
    ``hyp`` Nhyp x B with B > 1 evaluates all B vectors (the walkers / restarts of gplite_train.m:181,251,292)
    in ONE batched device pass and returns nlZ (B) and dnlZ (Nhyp x B); the reference raises
    gplite_nlZ:NoSampling for that form when a gradient is requested (:41-44).
    """
    engine = engine or default_engine()
    ctx = engine.ctx
    X = f64(gp["X"])
    N, D = X.shape
    y = f64(np.asarray(gp["y"], dtype=np.float64).reshape(-1))
    s2 = gp.get("s2")
    s2 = None if s2 is None or np.size(s2) == 0 else f64(np.asarray(s2, dtype=np.float64).reshape(-1))
    H = np.asarray(hyp, dtype=np.float64)
    single = H.ndim == 1 or H.shape[1] == 1
    H = f64(H.reshape(H.shape[0], -1))
    Nhyp, B = H.shape
    noisefun = tuple(gp["noisefun"])
    if Nhyp != gp["Ncov"] + gp["Nnoise"] + gp["Nmean"]:
        raise ValueError("gplite_nlZ:dimmismatch Number of hyperparameters mismatched with dimension of training inputs.")
    if gp.get("intmeanfun", 0) or gp.get("outwarpfun") is not None or int(np.atleast_1d(gp.get("covfun", 1))[0]) != 1:
        from ._lib import VBMC_ERR_UNSUPPORTED, VbmcUnsupported
        raise VbmcUnsupported(VBMC_ERR_UNSUPPORTED, "gplite_nlZ: integrated mean / output warping / non-SE covariance are not accelerated")
    nf = (C.c_int32 * 3)(*[int(v) for v in (list(noisefun) + [0, 0, 0])[:3]])
    grad = nargout > 1
    nlZ = np.zeros(B)
    dnlZ = np.zeros((Nhyp, B), order="F") if grad else None
    ctx.check(ctx.lib.vbmc_gp_nlz(ctx.h, N, D, B, Nhyp, int(gp["meanfun"]), nf, ptr(X), ptr(y), ptr(s2), ptr(H), int(grad),
                                  ptr(nlZ), ptr(dnlZ)))
    if hprior is not None:
        for b in range(B):
            P, dP = gplite_hypprior(H[:, b], hprior)
            nlZ[b] -= P
            if grad:
                dnlZ[:, b] -= dP
    if single:
        return (float(nlZ[0]), dnlZ[:, 0].copy()) if grad else float(nlZ[0])
    return (nlZ, dnlZ) if grad else nlZ
