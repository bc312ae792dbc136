"""Would two streams help the pipelined step at small batches?  Two contexts on the same device (each its own stream and scratch),
steps alternating between them, against one context with two slots (GPU box).  usage: python tools/r4_two_ctx.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402

D, N, K, S, Ns = 10, 400, 50, 20, 10000
inp = synth_inputs(0, D, N, K, S)
engs = [vbmc_amd.Engine(0), vbmc_amd.Engine(0)]
vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
theta0 = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
gps = [vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=e) for e in engs]
for R in (64, 16, 8, 4):
    th = np.asfortranarray(theta0[:, None] + 0.05 * np.random.default_rng(100).standard_normal((theta0.size, R)))
    objs = [vbmc_amd.PreparedObjective(theta0.size, R, 0, vp, gps[i], Ns, 0, None, engine=engs[i]) for i in range(2)]
    n = 40 if R >= 32 else 120

    def one_ctx():
        for _ in objs[0].stream([th] * n, seeds=list(range(n))):
            pass

    def two_ctx(depth):
        pend = []
        for i in range(n):
            o = objs[i & 1]
            sl = (i >> 1) % depth
            o.submit(th, seed=i, slot=sl)
            pend.append((o, sl))
            if len(pend) == 2 * depth:
                po, ps = pend.pop(0)
                po.collect(ps)
        while pend:
            po, ps = pend.pop(0)
            po.collect(ps)

    res = {}
    for name, fn in (("one context, two slots", one_ctx), ("two contexts x one slot", lambda: two_ctx(1)), ("two contexts x two slots", lambda: two_ctx(2))):
        fn()
        ts = []
        for rep in range(3):
            t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) / n)
        res[name] = 1e3 * float(np.median(ts))
    print("R = %d: " % R + "; ".join("%s %.4f ms" % kv for kv in res.items()))
