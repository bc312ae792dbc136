"""Acquisition sweep on the GPU behind the reference's call surface.

``acqwrapper_vbmc(Xs, vp, gp, optimState, transpose_flag, acqFun, acqInfo)`` (acq/acqwrapper_vbmc.m:1) with the
density-based acquisition functions ``acqf_vbmc`` (default, vbmc.m:213), ``acqflog_vbmc``, ``acqus_vbmc``,
``acqfsn2_vbmc`` and the importance-sampled ``acqviqr_vbmc`` / ``acqimiqr_vbmc`` (noisy targets): GP prediction for every hyper-sample, the hyper-sample statistics, the variational-posterior
density and the acquisition value are one fused device pass (``vbmc_acq_eval``).  The two steps that need VBMC's
variable transform -- the integer mapping (:8) and the hard-bound test in the ORIGINAL space (:49-51) -- are the
caller's: pass the boolean mask of out-of-bounds points as ``outside``.
"""
from __future__ import annotations

import numpy as np

from ._lib import VbmcUnsupported, f64, ptr
from .elbo import default_engine
from .gplite import _device_gp_with_noise

ACQ_IDS = {"acqf_vbmc": 0, "acqflog_vbmc": 1, "acqus_vbmc": 2, "acqfsn2_vbmc": 3, "acqviqr_vbmc": 10, "acqimiqr_vbmc": 11}


class ImportanceState:
    """Device copy of optimState.ActiveImportanceSampling (vbmc_acq_is_create); freed with the object."""

    def __init__(self, engine, dgp, ais):
        import ctypes as C

        self.ctx = engine.ctx
        Xa = np.asarray(ais["Xa"], dtype=np.float64)
        per_s = Xa.ndim == 3
        Na = Xa.shape[0]
        xa = f64(Xa.reshape(Na, -1, order="F")) if per_s else f64(Xa)
        lnw = ais.get("lnw")
        lnw = None if lnw is None or np.size(lnw) == 0 else f64(np.asarray(lnw, dtype=np.float64).reshape(dgp.S, Na))
        fs2a = ais.get("fs2a")
        fs2a = None if fs2a is None else f64(np.asarray(fs2a, dtype=np.float64).reshape(Na, dgp.S))
        ct = ais.get("Ctmp_mat")
        ct = None if ct is None else f64(np.asarray(ct, dtype=np.float64).reshape(dgp.N, -1, order="F"))
        self.h = C.c_void_p()
        self.ctx.check(self.ctx.lib.vbmc_acq_is_create(self.ctx.h, dgp.h, Na, ptr(xa), int(per_s), ptr(lnw), ptr(fs2a), ptr(ct),
                                                       C.byref(self.h)))

    def __del__(self):
        try:
            if self.h:
                self.ctx.lib.vbmc_acq_is_free(self.ctx.h, self.h)
                self.h = None
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


def _importance_state(engine, dgp, ais):
    """One-entry cache on the ActiveImportanceSampling dict itself (it is rebuilt once per active-sampling step,
    private/activesample_vbmc.m:209-212, and then reused by every acquisition call of that step)."""
    st = ais.get("_device")
    if st is None or st[0] is not dgp:
        st = (dgp, ImportanceState(engine, dgp, ais))
        ais["_device"] = st
    return st[1]


def acq_info(acqFun):
    """acqFun([]) info struct of the accelerated functions (acq/acqflog_vbmc.m:6-11)."""
    name = acqFun if isinstance(acqFun, str) else getattr(acqFun, "__name__", str(acqFun))
    name = name.lstrip("@")
    if name not in ACQ_IDS:
        raise VbmcUnsupported(-1, "acquisition function %s is not accelerated" % name)
    iqr = name in ("acqviqr_vbmc", "acqimiqr_vbmc")
    info = {"name": name, "log_flag": name == "acqflog_vbmc" or iqr, "compute_varlogjoint": False}
    if iqr:  # acq/acqviqr_vbmc.m:8-11, acq/acqimiqr_vbmc.m:8-10
        info.update(importance_sampling=True, importance_sampling_vp=False, variational_importance_sampling=name == "acqviqr_vbmc")
    return info


def acqwrapper_vbmc(Xs, vp, gp, optimState, transpose_flag=False, acqFun="acqf_vbmc", acqInfo=None, *, outside=None,
                    nargout=1, engine=None, shard=None):
    """acq = acqwrapper_vbmc(Xs,vp,gp,optimState,transpose_flag,acqFun,acqInfo).

    ``optimState`` keys used: ymax, VarianceRegularizedAcqFcn, TolGPVar (+ gplengthscale for acqfsn2 / acqviqr /
    acqimiqr, whose gp needs X_rescaled and sn2new, + ActiveImportanceSampling for the IQR functions).
    ``nargout=3`` also returns (fbar, vtot) of :21-29.

    Sharded form: ``shard`` = (rank, world, allgather) from vbmc_amd.dist.shard_spec() makes every rank evaluate the
    test points i = rank (mod world) on its own GPU (full GP replica) and all-gather the acquisition values, so that
    all ranks hold the identical vector and pick the identical argmin (private/activesample_vbmc.m:227-233)."""
    if shard is not None:
        rank, world, allgather = shard
        X_ = np.asarray(Xs, dtype=np.float64)
        if transpose_flag:
            X_ = X_.T
        X_ = X_.reshape(-1, gp["X"].shape[1])
        n = X_.shape[0]
        idx = np.arange(n)[rank::world]
        out_l = None if outside is None else np.asarray(outside, dtype=bool).reshape(-1)[idx]
        if idx.size:
            loc = _acq_local(X_[idx], vp, gp, optimState, acqFun, out_l, 3, engine)
        else:
            loc = (np.zeros(0), np.zeros(0), np.zeros(0))
        full = [allgather(v, idx, n) for v in (loc if nargout >= 3 else loc[:1])]
        acq = full[0].reshape(1, -1) if transpose_flag else full[0]
        return (acq, full[1], full[2]) if nargout >= 3 else acq
    return _acq_local(Xs, vp, gp, optimState, acqFun, outside, nargout, engine, transpose_flag)


def _acq_local(Xs, vp, gp, optimState, acqFun, outside, nargout, engine, transpose_flag=False):
    """The single-GPU evaluation behind acqwrapper_vbmc (one fused device pass)."""
    engine = engine or default_engine()
    ctx = engine.ctx
    info = acq_info(acqFun)
    acq_id = ACQ_IDS[info["name"]]
    delta = vp.get("delta")
    if delta is not None and np.any(np.asarray(delta) > 0):
        raise VbmcUnsupported(-1, "vp.delta > 0 (gplite_quad, acqwrapper_vbmc.m:12-14) is not accelerated")
    Xs = np.asarray(Xs, dtype=np.float64)
    if transpose_flag:
        Xs = Xs.T
    Xs = f64(Xs.reshape(-1, gp["X"].shape[1]))
    Nstar, D = Xs.shape
    K = int(vp["K"])
    dgp = _device_gp_with_noise(engine, gp)
    mu = f64(np.asarray(vp["mu"], dtype=np.float64).reshape(D, K))
    sigma = f64(np.asarray(vp["sigma"], dtype=np.float64).reshape(K))
    lam = f64(np.asarray(vp["lambda"], dtype=np.float64).reshape(D))
    w = f64(np.asarray(vp["w"], dtype=np.float64).reshape(K))
    gl = xr = sn = None
    if acq_id >= 10:
        gl = f64(np.asarray(optimState["gplengthscale"], dtype=np.float64).reshape(D))
        xr = f64(np.asarray(gp["X_rescaled"], dtype=np.float64))
        sn = f64(np.asarray(gp["sn2new"], dtype=np.float64).reshape(-1))
        ist = _importance_state(engine, dgp, optimState["ActiveImportanceSampling"])
        acq = np.zeros(Nstar)
        fbar = np.zeros(Nstar)
        vtot = np.zeros(Nstar)
        ctx.check(ctx.lib.vbmc_acq_iqr_eval(ctx.h, dgp.h, ist.h, Nstar, ptr(Xs), ptr(gl), ptr(xr), ptr(sn),
                                            int(bool(optimState.get("VarianceRegularizedAcqFcn", False))),
                                            float(optimState.get("TolGPVar", 0.0)), ptr(acq), ptr(fbar), ptr(vtot)))
        if outside is not None:
            acq = np.where(np.asarray(outside, dtype=bool).reshape(-1), np.inf, acq)
        if transpose_flag:
            acq = acq.reshape(1, -1)
        return (acq, fbar, vtot) if nargout >= 3 else acq
    if acq_id == 3:
        gl = f64(np.asarray(optimState["gplengthscale"], dtype=np.float64).reshape(D))
        xr = f64(np.asarray(gp["X_rescaled"], dtype=np.float64))
        sn = f64(np.asarray(gp["sn2new"], dtype=np.float64).reshape(-1))
    acq = np.zeros(Nstar)
    fbar = np.zeros(Nstar)
    vtot = np.zeros(Nstar)
    ctx.check(ctx.lib.vbmc_acq_eval(ctx.h, dgp.h, Nstar, ptr(Xs), acq_id, K, ptr(mu), ptr(sigma), ptr(lam), ptr(w),
                                    float(optimState.get("ymax", 0.0)), int(bool(optimState.get("VarianceRegularizedAcqFcn", False))),
                                    float(optimState.get("TolGPVar", 0.0)), ptr(gl), ptr(xr), ptr(sn), ptr(acq), ptr(fbar), ptr(vtot)))
    if outside is not None:
        acq = np.where(np.asarray(outside, dtype=bool).reshape(-1), np.inf, acq)   # :49-51
    if transpose_flag:
        acq = acq.reshape(1, -1)
    return (acq, fbar, vtot) if nargout >= 3 else acq


def vbmc_rnd(vp, N, origflag=False, balanceflag=False, *, rng=None):
    """[X,I] = vbmc_rnd(vp,N,0,balanceflag)  (vbmc_rnd.m:50-109): N draws from the Gaussian-mixture variational
    posterior in the TRANSFORMED space (origflag must be false here: the inverse variable transform is VBMC's
    warpvars_vbmc).  Host-side; used to place the importance points of the IQR acquisition functions."""
    if origflag:
        raise VbmcUnsupported(-1, "vbmc_rnd: origflag = 1 needs warpvars_vbmc (caller's side)")
    rng = np.random.default_rng() if rng is None else rng
    D, K = int(vp["D"]), int(vp["K"])
    N = int(N)
    if N < 1:
        return np.zeros((0, D)), np.zeros(0, dtype=int)
    w = np.asarray(vp["w"], dtype=np.float64).reshape(K)
    mu_t = np.asarray(vp["mu"], dtype=np.float64).reshape(D, K).T
    sigma = np.asarray(vp["sigma"], dtype=np.float64).reshape(K)
    lam = np.asarray(vp["lambda"], dtype=np.float64).reshape(D)
    if K > 1:
        if balanceflag:                                       # exact split by weight + weighted remainder (:69-88)
            n_floor = np.floor(w * N).astype(int)
            I = np.repeat(np.arange(K), n_floor)
            if N > I.size:
                w_extra = w * N - n_floor
                n_extra = int(np.ceil(np.sum(w_extra)))
                w_extra = w_extra + w * (n_extra - np.sum(w_extra))
                I = np.concatenate([I, rng.choice(K, size=n_extra, p=w_extra / np.sum(w_extra))])
            I = I[rng.permutation(I.size)[:N]]
        else:
            I = rng.choice(K, size=N, p=w / np.sum(w))        # catrnd(w,N) (:90)
        X = mu_t[I] + lam[None, :] * (rng.standard_normal((N, D)) * sigma[I][:, None])      # :95
    else:
        I = np.zeros(N, dtype=int)
        X = mu_t + lam[None, :] * (rng.standard_normal((N, D)) * sigma[0])                   # :103
    return X, I


def activeimportancesampling_vbmc(vp, gp, acqFun, acqInfo=None, options=None, *, rng=None, engine=None):
    """ActiveImportanceSampling = activeimportancesampling_vbmc(vp,gp,acqfun,acqinfo,options) for the branch VBMC takes
    with acqviqr_vbmc (variational_importance_sampling: private/activeimportancesampling_vbmc.m:36-52,92-100 and the
    Step-3 precomputation :248-276): Na draws from the variational posterior, lnw = 0, and -- on the device, inside
    vbmc_acq_is_create at the first acquisition call -- fs2a = gplite_pred at the draws and Ctmp = (L\\(L'\\Kax'))/sn2_eff.
    The MCMC refinement (:54-90, only for acqimiqr-style functions or a tiny effective sample size) is not mirrored."""
    info = acqInfo or acq_info(acqFun)
    if not info.get("variational_importance_sampling", False):
        raise VbmcUnsupported(-1, "activeimportancesampling_vbmc: only the variational (VIQR) branch is mirrored")
    opts = dict(options or {})
    na = opts.get("ActiveImportanceSamplingMCMCSamples", 100)          # vbmc.m:337 default '100'
    Na = int(np.ceil(na(int(vp["K"]), int(vp["D"])) if callable(na) else na))
    if Na <= 0:
        raise ValueError("OPTIONS.ActiveImportanceSamplingMCMCSamples should be (or evaluate to) a positive integer.")
    Xa, _ = vbmc_rnd(vp, Na, False, rng=rng)
    S = len(gp["post"])
    return {"Xa": Xa, "lnw": np.zeros((S, Na))}        # fs2a / Ctmp_mat are produced on the device from Xa
