"""Timing of the non-headline device paths at the C3 shape: gplite_post, gplite_pred (2^13 points), full-variance ELCBO,
gplite_nlZ + gradient batched over hyper-parameter vectors (with a LAPACK-backed NumPy evaluation of the same objective timed beside it)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402

D, N, K, S = 10, 400, 50, 20
inp = synth_inputs(0, D, N, K, S)
eng = vbmc_amd.Engine(0)


def timeit(f, n=5, warm=1):
    for _ in range(warm):
        f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n


out = {}
# every call leaves a new surrogate in the engine's cache (4 kept): the block pool reaches its steady state after ~6 calls
out["gplite_post_ms"] = 1e3 * timeit(lambda: vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng), 5, warm=6)
# the same with the N x N x S factors left on the device (no 25.6 MB readback): what the accelerated consumers need
out["gplite_post_resident_ms"] = 1e3 * timeit(lambda: vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None,
                                                                           need_L=False, engine=eng), 5, warm=6)
gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng)
Xs = 1.5 * np.random.default_rng(0).standard_normal((8192, D))
# (three warm-up calls: the first two after a posterior that downloaded its factor grow the pools and touch the result pages -- 9-10 ms each,
# profiles/r06_aux.md section 1; with one warm-up the mean of three read 3.8 ms)
out["gplite_pred_8192_ms"] = 1e3 * timeit(lambda: vbmc_amd.gplite_pred(gp, Xs, None, None, False, engine=eng), 5, warm=3)
vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
theta = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
out["eval_fullelcbo_ms"] = 1e3 * timeit(lambda: vbmc_amd.negelcbo_vbmc(theta, 0, vp, gp, 4096, 0, 1, nargout=11, engine=eng), 5)
out["diagvar_grad_ms"] = 1e3 * timeit(lambda: vbmc_amd.negelcbo_vbmc(theta, 1.0, vp, gp, 128, 1, 2, nargout=2, engine=eng), 5)
out["entlb_sieve_R250_ms"] = 1e3 * timeit(lambda: vbmc_amd.negelcbo_batch(np.tile(theta[:, None], (1, 250)), 0, vp, gp, 0, False, 0, engine=eng), 5)
st = {"ymax": float(np.max(inp["y"])), "VarianceRegularizedAcqFcn": True, "TolGPVar": 1e-4}
out["acqwrapper_acqf_8192_ms"] = 1e3 * timeit(lambda: vbmc_amd.acqwrapper_vbmc(Xs, vp, gp, st, False, "acqf_vbmc", None, engine=eng), 3)
gl = np.exp(np.mean(inp["hyp"][:D], axis=1))
gpn = dict(gp, X_rescaled=inp["X"] / gl[None, :], sn2new=np.full(N, 0.05))
Xa = 1.2 * np.random.default_rng(2).standard_normal((100, D))
stv = dict(st, gplengthscale=gl, ActiveImportanceSampling={"Xa": Xa})
out["acqwrapper_acqviqr_8192_Na100_ms"] = 1e3 * timeit(lambda: vbmc_amd.acqwrapper_vbmc(Xs, vp, gpn, stv, False, "acqviqr_vbmc", None, engine=eng), 3)
out["acqviqr_mfma_gflop"] = 2.0 * 8192 * 112 * N * S / 1e9
# GP hyper-parameter objective: B walkers, value + gradient
gpd = {"X": inp["X"], "y": inp["y"], "s2": None, "covfun": 1, "Ncov": D + 1, "noisefun": (1, 0, 0), "Nnoise": 1, "meanfun": 4,
       "Nmean": 2 * D + 1, "intmeanfun": 0}
for B in (1, 16, 64, 256):
    H = np.tile(inp["hyp"], (1, (B + S - 1) // S))[:, :B] + 0.01 * np.random.default_rng(1).standard_normal((inp["hyp"].shape[0], B))
    ms = 1e3 * timeit(lambda: vbmc_amd.gplite_nlZ(H, gpd, engine=eng), 3)
    out["nlz_grad_B%d_ms" % B] = ms
    out["nlz_grad_B%d_evals_per_s" % B] = B / (ms * 1e-3)
    msv = 1e3 * timeit(lambda: vbmc_amd.gplite_nlZ(H, gpd, nargout=1, engine=eng), 3)
    out["nlz_value_B%d_evals_per_s" % B] = B / (msv * 1e-3)
try:
    import scipy.linalg as sla

    def cpu_nlz(h):                     # LAPACK-backed equivalent of gplite_core.m:52-102,205,240-274 for a fair CPU number
        ell = np.exp(h[:D]); sf2 = np.exp(2 * h[D]); sn2 = np.exp(2 * h[D + 1])
        Xs_ = inp["X"] / ell
        d2 = np.maximum(np.sum(Xs_**2, 1)[:, None] + np.sum(Xs_**2, 1)[None, :] - 2 * Xs_ @ Xs_.T, 0)
        Km = sf2 * np.exp(-d2 / 2)
        Lc = sla.cholesky(Km / sn2 + np.eye(N))
        hm = h[D + 2:]
        m = hm[0] - 0.5 * np.sum(((inp["X"] - hm[1:D + 1]) / np.exp(hm[D + 1:2 * D + 1])) ** 2, axis=1)   # negquad mean (gplite_meanfun.m:425-431)
        al = sla.cho_solve((Lc, False), inp["y"] - m) / sn2
        Q = sla.cho_solve((Lc, False), np.eye(N)) / sn2 - np.outer(al, al)
        g = [np.sum(Q * Km * (Xs_[:, i][:, None] - Xs_[:, i][None, :]) ** 2) / 2 for i in range(D)]
        return (inp["y"] - m) @ al / 2 + np.sum(np.log(np.diag(Lc))), g, np.sum(Q * Km), np.trace(Q)
    # BLAS team sized to the cores the process is granted (cgroup quota), not to the logical CPUs it sees
    ncore = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            ncore = min(ncore, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        pass
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=ncore):
            t = timeit(lambda: cpu_nlz(inp["hyp"][:, 0]), 5)
    except ImportError:
        t = timeit(lambda: cpu_nlz(inp["hyp"][:, 0]), 5)
    out["nlz_grad_cpu_lapack_evals_per_s"] = 1.0 / t
    out["nlz_grad_cpu_lapack_cores"] = ncore
except Exception as e:  # noqa: BLE001
    out["nlz_cpu_error"] = repr(e)
print(json.dumps(out))
