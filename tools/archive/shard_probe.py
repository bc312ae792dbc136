"""Per-rank critical path of the hyper-sample / sample-chunk sharded evaluation, measured in ONE process on one GPU:
t(begin of rank 0 of W) + t(finish), against the unsharded pass.  The all-gather itself (RCCL over xGMI, ~0.3 MB per
rank at the headline shape) is not included -- there is one GPU here.  Usage: python tools/shard_probe.py [R] [Ns] [D N K S]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import vbmc_amd as va  # noqa: E402
from bench import synth_inputs  # noqa: E402
from vbmc_amd import elbo as E  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1
Ns = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
D, N, K, S = (int(v) for v in sys.argv[3:7]) if len(sys.argv) > 6 else (10, 400, 50, 20)
inp = synth_inputs(0, D, N, K, S)
eng = va.Engine(0)
gp = va.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng)
vp = va.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
theta0 = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
th = E.f64(theta0[:, None] + 0.05 * np.random.default_rng(1).standard_normal((theta0.size, R)))
ctx = eng.ctx
dgp = eng.device_gp(gp)
a, keep, _ = E._build_args(th, 0.0, vp, gp, Ns, True, 0, None, False, None, None, False, 3, eng)
F = np.empty(R)
dF = np.empty((th.shape[0], R), order="F")
a.F, a.dF = E.ptr(F), E.ptr(dF)


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t) / n * 1e6


base = timeit(lambda: ctx.check(ctx.lib.vbmc_elbo_batch(ctx.h, dgp.h, C.byref(a))))
print("D=%d N=%d K=%d S=%d unsharded R=%d Ns=%d: %.1f us per pass" % (D, N, K, S, R, Ns, base))
for W in (2, 4, 8):
    n = C.c_size_t(0)
    ctx.check(ctx.lib.vbmc_elbo_shard_size(ctx.h, dgp.h, C.byref(a), W, C.byref(n)))
    buf = torch.zeros(n.value * W, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for g in range(W):
        ctx.check(ctx.lib.vbmc_elbo_shard_begin(ctx.h, dgp.h, C.byref(a), g, W, C.c_void_p(buf.data_ptr() + 8 * n.value * g)))
    tb = timeit(lambda: ctx.check(ctx.lib.vbmc_elbo_shard_begin(ctx.h, dgp.h, C.byref(a), 0, W, C.c_void_p(buf.data_ptr()))))
    tf = timeit(lambda: ctx.check(ctx.lib.vbmc_elbo_shard_finish(ctx.h, dgp.h, C.byref(a), W, C.c_void_p(buf.data_ptr()))))
    a.chunk_world = 0
    print("world %d: block %.0f KB per rank; begin %.1f us + finish %.1f us = %.1f us (+ all-gather) -> x%.2f of the unsharded pass"
          % (W, n.value * 8 / 1024, tb, tf, tb + tf, base / (tb + tf)))
