"""negelcbo_vbmc on the GPU: the reference's objective call surface over the C ABI.

``negelcbo_vbmc(theta, beta, vp, gp, Ns, compute_grad, compute_var, altent_flag, thetabnd,
entropy_alpha)`` keeps the reference's positional arguments (misc/negelcbo_vbmc.m:1) and
returns the reference's 11 outputs as a tuple truncated to ``nargout``.  ``negelcbo_batch``
evaluates R thetas in one launch (sieve batch / many Adam chains).

``gp`` is a plain dict with the reference's field names: X, y, s2, covfun, meanfun, noisefun,
Ncov, Nnoise, Nmean, post = [ {hyp, alpha, sW, L, sn2_mult, Lchol}, ... ] (gplite_post.m:94-157).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import Context, DeviceGP, ElboArgs, VbmcUnsupported, f64, ptr
from .vp import get_vptheta

_engines = {}


class Engine:
    """A device context plus a cache of uploaded GP posteriors."""

    def __init__(self, device=0, stream=None):
        self.ctx = Context(device, stream)
        self._gp_cache = {}

    def device_gp(self, gp, need_L=False):
        # keyed on the identity of the posterior list: shallow copies of the gp dict that only add fields (X_rescaled, sn2new
        # for the acquisition functions) share the uploaded posterior
        post = gp["post"]
        key = id(post)
        ent = self._gp_cache.get(key)
        if ent is not None and ent[0] is post and (ent[2] or not need_L) and ent[3] == self._fingerprint(gp):
            self._gp_cache[key] = self._gp_cache.pop(key)   # most recently used last
            return ent[1]
        S = len(post)
        X = np.asarray(gp["X"], dtype=np.float64)
        N = X.shape[0]
        hyp = np.stack([np.asarray(p["hyp"], dtype=np.float64).reshape(-1) for p in post], axis=1)
        alpha = np.stack([np.asarray(p["alpha"], dtype=np.float64).reshape(-1) for p in post], axis=1)
        L = None
        if need_L:
            if any(p["L"] is None for p in post):
                raise ValueError("this gp was produced with need_L=False (factors kept on the device only) and its device copy "
                                 "is no longer cached: rebuild it with gplite_post or request need_L=True")
            L = np.stack([np.asarray(p["L"], dtype=np.float64) for p in post], axis=2)
        sW1 = np.array([np.asarray(p["sW"]).reshape(-1)[0] for p in post])
        lch = np.array([1 if p["Lchol"] else 0 for p in post], dtype=np.uint8)
        dgp = DeviceGP(self.ctx, X, hyp, alpha, L, sW1, lch, gp["meanfun"], gp["Ncov"], gp["Nnoise"])
        self._remember(gp, dgp, need_L)
        return dgp

    def _remember(self, gp, dgp, has_L):
        """Keep the device copies of the last few surrogates (the current one and the ones it was appended from / to):
        a few tens of MB each against 288 GB of HBM."""
        self._gp_cache.pop(id(gp["post"]), None)
        self._gp_cache[id(gp["post"])] = (gp["post"], dgp, has_L, self._fingerprint(gp))
        while len(self._gp_cache) > 4:
            self._gp_cache.pop(next(iter(self._gp_cache)))

    @staticmethod
    def _fingerprint(gp):
        p0 = gp["post"][0]
        return (id(gp["X"]), np.asarray(gp["X"]).shape, len(gp["post"]), int(gp["meanfun"]), int(gp["Ncov"]), int(gp["Nnoise"]),
                float(np.asarray(p0["alpha"]).reshape(-1)[0]), float(np.asarray(p0["hyp"]).reshape(-1)[0]))

    def invalidate(self):
        self._gp_cache = {}


def default_engine(device=0):
    eng = _engines.get(device)
    if eng is None:
        eng = Engine(device)
        _engines[device] = eng
    return eng


def _flags(vp):
    return (int(bool(vp["optimize_mu"])), int(bool(vp["optimize_sigma"])), int(bool(vp["optimize_lambda"])),
            int(bool(vp["optimize_weights"])))


def _build_args(thetas, beta, vp, gp, Ns, compute_grad, compute_var, thetabnd, separate_K, eps, eps_device_ptr, eps_shared,
                seed, engine, sparse_cutoff=0.0, chunk_world=0):
    """Fill a vbmc_elbo_args for R = thetas.shape[1] restarts; returns (args, keep-alive list, compute_var)."""
    D, K = int(vp["D"]), int(vp["K"])
    T, R = thetas.shape
    if beta is None or not np.isfinite(beta):
        beta = 0.0  # negelcbo_vbmc.m:15
    if compute_var is None:
        compute_var = 1 if beta != 0 else 0  # :16 (first clause; nargout clause handled by callers)
    compute_var = int(compute_var)
    a = ElboArgs()
    a.struct_size = C.sizeof(ElboArgs)
    a.D, a.K, a.R = D, K, R
    fl = _flags(vp)
    for i in range(4):
        a.optimize[i] = fl[i]
    keep = [thetas]
    a.theta = ptr(thetas)

    def hold(x):
        x = f64(x)
        keep.append(x)
        return ptr(x)

    a.vp_mu = hold(vp["mu"])
    a.vp_sigma = hold(vp["sigma"])
    a.vp_lambda = hold(vp["lambda"])
    a.vp_w = hold(vp["w"])
    delta = vp.get("delta")
    if delta is not None and np.size(delta) > 0 and np.any(np.asarray(delta) != 0):
        a.vp_delta = hold(np.broadcast_to(np.asarray(delta, dtype=np.float64).reshape(-1), (D,)).copy())
    Ns = int(Ns)
    a.Ns = Ns
    a.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    a.eps_shared = 1 if eps_shared else 0
    if Ns > 0 and eps_device_ptr is not None:
        a.eps_mode = 2
        a.eps = C.c_void_p(int(eps_device_ptr))
    elif Ns > 0 and eps is not None:
        Mh = (Ns + 1) // 2
        e = np.ascontiguousarray(np.asarray(eps, dtype=np.float64))
        want_shape = (K, Mh, D) if eps_shared else (R, K, Mh, D)
        if e.shape != want_shape and not (R == 1 and e.shape == (K, Mh, D)):
            raise ValueError("eps has shape %r, expected %r" % (e.shape, want_shape))
        keep.append(e)
        a.eps_mode = 1
        a.eps = C.c_void_p(e.ctypes.data)
    else:
        a.eps_mode = 0
    a.compute_grad = 1 if compute_grad else 0
    a.compute_var = compute_var
    a.separate_K = 1 if separate_K else 0
    a.beta = float(beta)
    a.sparse_cutoff = float(sparse_cutoff or 0.0)
    a.chunk_world = int(chunk_world or 0)
    if thetabnd is not None:
        a.bnd_lb = hold(thetabnd["lb"])
        a.bnd_ub = hold(thetabnd["ub"])
        a.TolCon = float(thetabnd["TolCon"])
        a.WeightThreshold = float(thetabnd.get("WeightThreshold", 0.0))
        a.WeightPenalty = float(thetabnd.get("WeightPenalty", 0.0))
    return a, keep, compute_var


def fminadam_device(x0, beta, vp, gp, Ns, thetabnd=None, TolFun=1e-3, MaxIter=10000, master_stepsize=None, *,
                    compute_var=0, seed=0, engine=None, sparse_cutoff=0.0, tables=True):
    """fminadam (utils/fminadam.m) with the objective negelcbo_vbmc(., beta, vp, gp, Ns, 1, compute_var, ~, thetabnd)
    run entirely on the device for R chains in lock-step (x0: (T, R) or (T,)).

    Returns (x, f, xtab, ftab, iters): x (T, R) mean of the last 20 iterates, f (R,), xtab list of (T, iters_r)
    arrays, ftab list of (iters_r,) arrays, iters (R,) -- the reference's five outputs per chain.
    tables=False: the iterate tables stay on the device; (x, f, xmid, None, iters) with xmid (T, R) = each chain's iterate of
    smallest recorded objective (what misc/vpoptimize_vbmc.m:133 reads from the tables)."""
    engine = engine or default_engine()
    ctx = engine.ctx
    ms = {"max": 0.1, "min": 0.001, "decay": 200.0}
    if master_stepsize:
        ms.update({k: v for k, v in master_stepsize.items() if v is not None})
    x0 = f64(x0)
    if x0.ndim == 1:
        x0 = f64(x0.reshape(-1, 1))
    T, R = x0.shape
    a, keep, _ = _build_args(x0, beta, vp, gp, Ns, True, compute_var, thetabnd, False, None, None, False, seed, engine,
                             sparse_cutoff)
    MaxIter = int(MaxIter)
    x = np.zeros((T, R), order="F")
    f = np.zeros(R)
    iters = np.zeros(R, dtype=np.int32)
    xtab = np.zeros((R, MaxIter, T)) if tables else None   # C order == T x MaxIter x R column-major
    ftab = np.zeros((R, MaxIter)) if tables else None
    xmid = np.zeros((T, R), order="F")
    dgp = engine.device_gp(gp, need_L=int(compute_var or 0) != 0)
    ctx.check(ctx.lib.vbmc_adam_batch(ctx.h, dgp.h, C.byref(a), float(TolFun), MaxIter, float(ms["min"]), float(ms["max"]),
                                       float(ms["decay"]), ptr(x), ptr(f), iters.ctypes.data_as(C.POINTER(C.c_int32)),
                                       ptr(xtab), ptr(ftab), ptr(xmid)))
    if not tables:
        return x, f, xmid, None, iters
    return x, f, [xtab[r, : iters[r]].T.copy() for r in range(R)], [ftab[r, : iters[r]].copy() for r in range(R)], iters


def negelcbo_batch(thetas, beta, vp, gp, Ns=0, compute_grad=True, compute_var=None, thetabnd=None, *,
                   separate_K=False, eps=None, eps_device_ptr=None, eps_shared=False, seed=0, engine=None,
                   sparse_cutoff=0.0, outputs=None, chunk_world=0, jacobian_flag=True, restart_offset=0, restart_stride=0,
                   plan_restarts=0):
    """R evaluations of negelcbo_vbmc in one device pass.

    outputs: None = everything below; a subset such as ("F", "dF") -- what the optimiser loop reads,
    misc/vpoptimize_vbmc.m:71 -- leaves the other ABI output pointers NULL so that only the requested
    gradients cross PCIe.
    thetas: (T, R) column per restart (or (T,) for R = 1).  Returns a dict of arrays
    F[R], dF[T,R], G[R], H[R], dG[T,R], dH[T,R], varG[R], varGss[R], I_sk[S,K,R], J_sjk[S,K,K,R].
    eps: host array shaped (R, K, Ns/2, D) (or (K, Ns/2, D) with eps_shared) standing in for the
    reference's randn stream (entmc_vbmc.m:53); None -> device Philox stream keyed by ``seed``.
    sparse_cutoff: 0 dense; c > 0 skips 16-component tiles whose terms are provably < exp(-c) of q(x).
    chunk_world: W > 1 chunks the MC samples as negelcbo_shard does for a world of W ranks (its bit-exact 1-GPU reference).
    jacobian_flag=False: gradients with respect to sigma, lambda, w themselves (the JACOBIAN_FLAG = 0 form of the stand-alone
    functions; no soft bounds).  outputs may name "dvarG" (T, R) and "dvarG_s" (T, S, R) with compute_var = 2 and a gradient.
    restart_offset / restart_stride / plan_restarts: this call is a SHARE of a larger batch (vbmc_elbo_args): the device stream of
    column r is that of restart offset + r stride of the undivided batch, and with plan_restarts = its size the launch shapes are
    the undivided batch's too -- every column bit-identical to the one-device evaluation of the whole batch.
    """
    engine = engine or default_engine()
    ctx = engine.ctx
    thetas = f64(thetas)
    if thetas.ndim == 1:
        thetas = f64(thetas.reshape(-1, 1))
    T, R = thetas.shape
    D, K = int(vp["D"]), int(vp["K"])
    a, keep, compute_var = _build_args(thetas, beta, vp, gp, Ns, compute_grad, compute_var, thetabnd, separate_K, eps,
                                       eps_device_ptr, eps_shared, seed, engine, sparse_cutoff, chunk_world)
    a.no_jacobian = 0 if jacobian_flag else 1
    a.restart_offset, a.restart_stride, a.plan_restarts = int(restart_offset), int(restart_stride), int(plan_restarts)
    if gp is None:   # entropy only (entmc_vbmc / entlb_vbmc on their own): the ABI takes a NULL surrogate
        if compute_var or separate_K:
            raise ValueError("an entropy-only evaluation has no variance or per-component outputs")
        dgp_h, S = None, 0
    else:
        dgp = engine.device_gp(gp, need_L=compute_var != 0)
        dgp_h, S = dgp.h, dgp.S
    out = {}

    def outbuf(name, shape):
        if outputs is not None and name not in outputs:
            return None
        arr = np.empty(shape, dtype=np.float64, order="F")
        out[name] = arr
        return ptr(arr)

    a.F = outbuf("F", (R,))
    a.G = outbuf("G", (R,))
    a.H = outbuf("H", (R,))
    if compute_grad:
        a.dF = outbuf("dF", (T, R))
        a.dG = outbuf("dG", (T, R))
        a.dH = outbuf("dH", (T, R))
    a.varG = outbuf("varG", (R,))
    a.varGss = outbuf("varGss", (R,))
    if outputs is not None and "dvarG" in outputs and compute_grad and compute_var == 2:
        a.dvarG = outbuf("dvarG", (T, R))
    if separate_K:
        a.I_sk = outbuf("I_sk", (S, K, R))
        if compute_var:
            a.J_sjk = outbuf("J_sjk", (S, K, K, R))
    if outputs is not None and gp is not None:     # per-hyper-sample outputs only on request (gplogjoint avg_flag = 0)
        if "G_s" in outputs:
            a.G_s = outbuf("G_s", (S, R))
        if "varG_s" in outputs and compute_var:
            a.varG_s = outbuf("varG_s", (S, R))
        if "dG_s" in outputs and compute_grad:
            a.dG_s = outbuf("dG_s", (T, S, R))
        if "dvarG_s" in outputs and compute_grad and compute_var == 2:
            a.dvarG_s = outbuf("dvarG_s", (T, S, R))
    ctx.check(ctx.lib.vbmc_elbo_batch(ctx.h, dgp_h, C.byref(a)))
    return out


def negelcbo_shard(thetas, beta, vp, gp, Ns, compute_grad=True, thetabnd=None, *, rank, world, exchange, seed=0, engine=None,
                   outputs=("F", "dF")):
    """ONE negelcbo evaluation (or a batch with fewer restarts than GPUs) sharded over ``world`` ranks along the GP
    hyper-sample axis (misc/gplogjoint.m:98) and the Monte-Carlo sample chunks of the entropy (ent/entmc_vbmc.m:49-104);
    every rank returns the outputs of ``negelcbo_batch`` -- bit-identical to the 1-GPU evaluation with
    ``chunk_world=world`` (vbmc_elbo_shard_*: the samples are cut into ``world`` times as many chunks as one device needs).

    ``exchange`` (vbmc_amd.dist.ShardExchange or anything with the same two methods): ``send_buffer(n)`` -> device
    pointer of n doubles this rank fills; ``all_gather()`` -> device pointer of the world blocks in rank order.
    Value + gradient without variance, device RNG (the optimiser-loop call, misc/vpoptimize_vbmc.m:71)."""
    engine = engine or default_engine()
    ctx = engine.ctx
    thetas = f64(thetas)
    if thetas.ndim == 1:
        thetas = f64(thetas.reshape(-1, 1))
    T, R = thetas.shape
    a, keep, _ = _build_args(thetas, beta, vp, gp, Ns, compute_grad, 0, thetabnd, False, None, None, False, seed, engine)
    dgp = engine.device_gp(gp)
    out = {}
    for name, shape in (("F", (R,)), ("G", (R,)), ("H", (R,))) + ((("dF", (T, R)), ("dG", (T, R)), ("dH", (T, R))) if compute_grad else ()):
        if outputs is None or name in outputs:
            out[name] = np.empty(shape, dtype=np.float64, order="F")
            setattr(a, name, ptr(out[name]))
    n = C.c_size_t(0)
    ctx.check(ctx.lib.vbmc_elbo_shard_size(ctx.h, dgp.h, C.byref(a), int(world), C.byref(n)))
    send = exchange.send_buffer(int(n.value))
    ctx.check(ctx.lib.vbmc_elbo_shard_begin(ctx.h, dgp.h, C.byref(a), int(rank), int(world), C.c_void_p(int(send))))
    gathered = exchange.all_gather()
    ctx.check(ctx.lib.vbmc_elbo_shard_finish(ctx.h, dgp.h, C.byref(a), int(world), C.c_void_p(int(gathered))))
    return out


def _with_grad_groups(vp, grad_flags, nargout, jacobian_flag, who):
    """grad_flags defaulting of the reference (ent/entmc_vbmc.m:5-11, misc/gplogjoint.m:17-23) -> (vp whose optimize_*
    flags select exactly the groups a gradient is wanted for, theta, any gradient)."""
    if nargout < 2:
        grad_flags = False
    elif grad_flags is None or np.size(grad_flags) == 0:
        grad_flags = True
    gf = np.broadcast_to(np.asarray(grad_flags, dtype=bool).reshape(-1), (4,)) if np.size(grad_flags) == 1 \
        else np.asarray(grad_flags, dtype=bool).reshape(4)
    vpt = dict(vp)
    if gf.any():
        for name, f in zip(("optimize_mu", "optimize_sigma", "optimize_lambda", "optimize_weights"), gf):
            vpt[name] = bool(f)
    # get_vptheta rescales (sigma*nl, lambda/nl, misc/rescale_params.m:26-32): the NON-flagged groups are read from the
    # returned vp, so it must be the rescaled one -- otherwise an un-normalised vp.lambda makes theta and the fixed
    # sigma / lambda inconsistent by the factor nl
    theta, vpr = get_vptheta(vpt)
    return vpr, theta, bool(gf.any())


def entmc_vbmc(vp, Ns=10, grad_flags=None, jacobian_flag=True, nargout=2, *, eps=None, seed=0, engine=None):
    """[H,dH] = entmc_vbmc(vp,Ns,grad_flags,jacobian_flag)  (ent/entmc_vbmc.m:1): Monte Carlo entropy of the mixture on
    its own.  dH holds the flagged groups in theta order [mu(:); log sigma; log lambda; eta] (:110-125)."""
    if Ns is None:
        Ns = 10   # :4
    vpt, theta, g = _with_grad_groups(vp, grad_flags, nargout, jacobian_flag, "entmc_vbmc")
    r = negelcbo_batch(theta, 0.0, vpt, None, int(Ns), g, 0, None, eps=eps, seed=seed, engine=engine,
                       outputs=("H", "dH") if g else ("H",), jacobian_flag=bool(jacobian_flag))
    H = float(r["H"][0])
    return (H, r["dH"][:, 0].copy() if g else np.zeros(0)) if nargout > 1 else H


def entlb_vbmc(vp, grad_flags=None, jacobian_flag=True, nargout=2, *, engine=None):
    """[H,dH] = entlb_vbmc(vp,grad_flags,jacobian_flag)  (ent/entlb_vbmc.m:1): deterministic entropy lower bound."""
    vpt, theta, g = _with_grad_groups(vp, grad_flags, nargout, jacobian_flag, "entlb_vbmc")
    r = negelcbo_batch(theta, 0.0, vpt, None, 0, g, 0, None, engine=engine, outputs=("H", "dH") if g else ("H",),
                       jacobian_flag=bool(jacobian_flag))
    H = float(r["H"][0])
    return (H, r["dH"][:, 0].copy() if g else np.zeros(0)) if nargout > 1 else H


def gplogjoint(vp, gp, grad_flags=None, avg_flag=True, jacobian_flag=True, compute_var=None, separate_K=None, nargout=1, *,
               engine=None):
    """[F,dF,varF,dvarF,varss,I_sk,J_sjk] = gplogjoint(vp,gp,grad_flags,avg_flag,jacobian_flag,compute_var,separate_K)
    (misc/gplogjoint.m:1-30).  Accelerated call forms: averaged over the hyper-parameter samples (avg_flag = 1) with
    transformed gradients (jacobian_flag), and per-hyper-sample values without gradients (avg_flag = 0: F and varF are
    length-S vectors, varss = 0 -- the forms of private/activesample_vbmc.m:155 and misc/vpoptimizeweights_vbmc.m:42);
    untransformed gradients (jacobian_flag = 0: with respect to sigma, lambda and w themselves, :352-373 skipped), dvarF --
    the gradient of the diagonal variance (compute_var = 2, nargout >= 4; :375-413) -- and, round 4, per-hyper-sample gradients
    (avg_flag = 0 with grad_flags: dF is T x S, :411 skipped; vbmc_elbo_args.dG_s) and per-component outputs together with gradients
    (separate_K with grad_flags: two passes); round 5: dvarF with jacobian_flag = 0 (:375-396 skipped) and per hyper-sample (avg_flag = 0:
    T x S, :407-409 skipped; vbmc_elbo_args.dvarG_s) -- every call form of the reference's gplogjoint with mean functions 0 / 1 / 4."""
    if separate_K is None:
        separate_K = nargout > 5            # :13
    if compute_var is None:
        compute_var = nargout > 2           # :14
    compute_var = int(compute_var)
    vpt, theta, g = _with_grad_groups(vp, grad_flags, nargout, jacobian_flag, "gplogjoint")
    want_dvar = nargout > 3 and compute_var and g      # compute_vargrad (:27)
    if want_dvar:
        if compute_var != 2:                # :27-30
            raise ValueError("gplogjoint:FullVarianceGradient Computation of gradient of log joint variance is currently "
                             "available only for diagonal approximation of the variance.")
    want = ["G"] + (["dG"] if g else []) + (["varG", "varGss"] if compute_var else []) + (["dvarG"] if want_dvar else [])
    if not avg_flag:
        want += ["G_s"] + (["varG_s"] if compute_var else []) + (["dG_s"] if g else []) + (["dvarG_s"] if want_dvar else [])
    sep = None
    if separate_K and g:
        # per-component outputs TOGETHER with gradients (round 4): the objective's entry point refuses the combination as
        # negelcbo_vbmc.m:57-59 does, so the stand-alone form takes two passes -- I_sk / J_sjk do not depend on the gradient request
        sep = negelcbo_batch(theta, 0.0, vpt, gp, 0, False, compute_var, None, separate_K=True, engine=engine,
                             outputs=("I_sk",) + (("J_sjk",) if compute_var else ()), jacobian_flag=bool(jacobian_flag))
    elif separate_K:
        want += ["I_sk"] + (["J_sjk"] if compute_var else [])
    r = negelcbo_batch(theta, 0.0, vpt, gp, 0, g, compute_var, None, separate_K=bool(separate_K) and sep is None, engine=engine,
                       outputs=tuple(want), jacobian_flag=bool(jacobian_flag))
    if sep is not None:
        r.update(sep)
    if not avg_flag and r["G_s"].shape[0] > 1:     # :399: no averaging -> F, varF are 1 x S, dF is T x S; varss stays 0 (:398)
        outs = (r["G_s"][:, 0].copy(), r["dG_s"][:, :, 0].copy() if g else np.zeros(0), r["varG_s"][:, 0].copy() if compute_var else None,
                r["dvarG_s"][:, :, 0].copy() if want_dvar else None, 0.0,
                r["I_sk"][:, :, 0].copy() if separate_K else None,
                r["J_sjk"][:, :, :, 0].copy() if (separate_K and compute_var) else None)
        return outs[0] if nargout <= 1 else outs[:nargout]
    outs = (float(r["G"][0]), r["dG"][:, 0].copy() if g else np.zeros(0),
            float(r["varG"][0]) if compute_var else None, r["dvarG"][:, 0].copy() if want_dvar else None,
            (float(r["varGss"][0]) if avg_flag else 0.0) if compute_var else None,
            r["I_sk"][:, :, 0].copy() if separate_K else None,
            r["J_sjk"][:, :, :, 0].copy() if (separate_K and compute_var) else None)
    return outs[0] if nargout <= 1 else outs[:nargout]


class PreparedObjective:
    """The optimiser-loop objective @(theta) negelcbo_vbmc(theta,beta,vp,gp,Ns,1,compute_var,0,thetabnd) with everything
    that does not change between calls resolved once (misc/vpoptimize_vbmc.m:71 builds exactly such a closure): the
    argument struct, the fixed vp groups, bounds, the device GP and the output buffers.  A call copies the new theta
    (T x R) into place and runs one batched device pass; it returns views of the same (F, dF) buffers every time."""

    def __init__(self, T, R, beta, vp, gp, Ns, compute_var=0, thetabnd=None, *, engine=None, sparse_cutoff=0.0):
        self.engine = engine or default_engine()
        self.theta = np.zeros((T, R), order="F")
        self.args, self._keep, cv = _build_args(self.theta, beta, vp, gp, Ns, True, compute_var, thetabnd, False, None, None,
                                                False, 0, self.engine, sparse_cutoff)
        self.dgp = self.engine.device_gp(gp, need_L=cv != 0)
        self.F = np.empty(R)
        self.dF = np.empty((T, R), order="F")
        self.args.F = ptr(self.F)
        self.args.dF = ptr(self.dF)
        self._ref = C.byref(self.args)
        self._slots = None

    def __call__(self, thetas, seed=0):
        np.copyto(self.theta, np.asarray(thetas, dtype=np.float64).reshape(self.theta.shape, order="F"))
        self.args.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        ctx = self.engine.ctx
        ctx.check(ctx.lib.vbmc_elbo_batch(ctx.h, self.dgp.h, self._ref))
        return self.F, self.dF

    # ---- pipelined form (vbmc_elbo_submit / vbmc_elbo_collect): for streams of INDEPENDENT batches, e.g. the candidates of
    # the sieve (misc/vpsieve_vbmc.m:74-78); the host stages batch i + 1 while the device works on batch i
    def _slot(self, slot):
        if self._slots is None:
            self._slots = {}
        if slot not in self._slots:
            a = type(self.args).from_buffer_copy(self.args)      # same inputs, its own output buffers
            F, dF = np.empty_like(self.F), np.empty_like(self.dF)
            a.F, a.dF = ptr(F), ptr(dF)
            self._slots[slot] = (a, F, dF, C.byref(a))           # (the reference is built once: a submit is ~20 us of host time in all)
        return self._slots[slot]

    def submit(self, thetas, seed=0, slot=0):
        """Enqueue one batch in ``slot`` (0 .. 3) and return without waiting; theta is copied before the call returns."""
        a, _, _, ref = self._slot(slot)
        th = thetas if (type(thetas) is np.ndarray and thetas.dtype == np.float64 and thetas.shape == self.theta.shape) else \
            np.asarray(thetas, dtype=np.float64).reshape(self.theta.shape, order="F")
        np.copyto(self.theta, th)
        a.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        ctx = self.engine.ctx
        st = ctx.lib.vbmc_elbo_submit(ctx.h, self.dgp.h, ref, slot)
        if st:
            ctx.check(st)

    def collect(self, slot=0):
        """Wait for the batch submitted in ``slot``; returns views of that slot's (F, dF) buffers."""
        _, F, dF, ref = self._slot(slot)
        ctx = self.engine.ctx
        st = ctx.lib.vbmc_elbo_collect(ctx.h, ref, slot)
        if st:
            ctx.check(st)
        return F, dF

    def stream(self, batches, seeds=None):
        """Generator over an iterable of theta batches (T x R each): yields (F, dF) copies in order, four batches in flight (two per
        stream; two in all for the variance forms, which stay on the context's own stream)."""
        pending = []
        import os

        depth = 4 if int(self.args.compute_var) == 0 and os.environ.get("VBMC_SLOT_STREAMS") != "0" else 2
        try:
            for i, th in enumerate(batches):
                self.submit(th, seed=(seeds[i] if seeds is not None else i), slot=i % depth)
                pending.append(i % depth)
                if len(pending) == depth:
                    F, dF = self.collect(pending.pop(0))
                    yield F.copy(), dF.copy()
            while pending:
                F, dF = self.collect(pending.pop(0))
                yield F.copy(), dF.copy()
        finally:
            # an abandoned generator, or an exception between a submit and its collect: no pass stays in flight behind a slot nobody
            # will collect (vbmc_elbo_abandon waits for it and frees the slot)
            for sl in pending:
                self.abandon(sl)

    def abandon(self, slot=0):
        """Give ``slot`` back without its results (waits for the pass in flight, if any)."""
        ctx = self.engine.ctx
        st = ctx.lib.vbmc_elbo_abandon(ctx.h, int(slot))
        if st:
            ctx.check(st)


def negelcbo_vbmc(theta, beta, vp, gp, Ns=0, compute_grad=None, compute_var=None, altent_flag=False, thetabnd=None,
                  entropy_alpha=0, nargout=2, *, eps=None, seed=0, engine=None):
    """[F,dF,G,H,varF,dH,varGss,varG,varH,I_sk,J_sjk] = negelcbo_vbmc(...)  (misc/negelcbo_vbmc.m:1).

    ``nargout`` plays MATLAB's role: compute_grad defaults to nargout > 1 (:10), compute_var to
    beta ~= 0 || nargout > 4 (:16), separate_K = nargout > 9 (:17).  altent_flag and
    entropy_alpha are accepted and ignored, as in the reference (:19).
    """
    if Ns is None:
        Ns = 0
    if compute_grad is None:
        compute_grad = nargout > 1
    if beta is None or not np.isfinite(beta):
        beta = 0.0
    if compute_var is None:
        compute_var = (beta != 0) or nargout > 4
    separate_K = nargout > 9
    r = negelcbo_batch(np.asarray(theta, dtype=np.float64).reshape(-1), beta, vp, gp, Ns, bool(compute_grad),
                       int(compute_var), thetabnd, separate_K=separate_K, eps=eps, seed=seed, engine=engine)
    F = float(r["F"][0])
    dF = r["dF"][:, 0].copy() if compute_grad else np.zeros(0)
    G, H = float(r["G"][0]), float(r["H"][0])
    varH = 0.0
    varG = float(r["varG"][0]) if compute_var else 0.0
    varF = varG + varH if compute_var else 0.0
    dH = r["dH"][:, 0].copy() if compute_grad else np.zeros(0)
    varGss = float(r["varGss"][0]) if compute_var else 0.0
    I_sk = r["I_sk"][:, :, 0].copy() if separate_K else None
    J_sjk = r["J_sjk"][:, :, :, 0].copy() if (separate_K and compute_var) else None
    outs = (F, dF, G, H, varF, dH, varGss, varG, varH, I_sk, J_sjk)
    return outs[: max(1, nargout)] if nargout < 11 else outs
