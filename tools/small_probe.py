"""The small-work class on the GPU box: one shape, pipelined rate + entropy-kernel duration (kernel alone, HIP events), for the
environment the caller set (VBMC_ENT_KERNEL, VBMC_LJ_CO, VBMC_ENT_CHUNKS ...).  Usage: python tools/small_probe.py [D N K Ns S R]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402

D, N, K, Ns, S, R = (int(x) for x in (sys.argv[1:7] if len(sys.argv) >= 7 else (6, 200, 10, 1000, 8, 64)))
eng = vbmc_amd.default_engine()
inp = synth_inputs(0, D, N, K, S, "student" if D == 6 else "lumpy", False)
gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, need_L=False, engine=eng)
vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
th0 = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
th = np.asfortranarray(th0[:, None] + 0.05 * np.random.default_rng(100).standard_normal((th0.size, R)))
obj = vbmc_amd.PreparedObjective(th0.size, R, 0, vp, gp, Ns, 0, None, engine=eng)
nsteps = 200
for _ in obj.stream([th] * 20, seeds=list(range(20))):
    pass
best = 1e9
for rep in range(3):
    t1 = time.perf_counter()
    for F_, dF_ in obj.stream([th] * nsteps, seeds=list(range(10, 10 + nsteps))):
        pass
    best = min(best, time.perf_counter() - t1)
eng.ctx.set_profiling(2)
ems, ljs = [], []
for i in range(20):
    vbmc_amd.negelcbo_batch(th, 0, vp, gp, Ns, True, 0, seed=50 + i, engine=eng, outputs=("F", "dF"))
    a, b = eng.ctx.last_kernel_ms()
    ems.append(a)
    ljs.append(b)
eng.ctx.set_profiling(False)
t1 = time.perf_counter()
for i in range(50):
    vbmc_amd.negelcbo_batch(th, 0, vp, gp, Ns, True, 0, seed=50 + i, engine=eng, outputs=("F", "dF"))
blk = (time.perf_counter() - t1) / 50
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("VBMC_")}, "shape": [D, N, K, Ns, S, R],
                  "evals_per_s": R * nsteps / best, "us_per_step": 1e6 * best / nsteps, "ent_kernel_us_med_min": [1e3 * float(np.median(ems)), 1e3 * float(np.min(ems))],
                  "lj_kernel_us": 1e3 * float(np.median(ljs)), "blocking_us": 1e6 * blk, "F0": float(F_[0]), "F_sum": float(np.sum(F_)), "dF_abs_sum": float(np.sum(np.abs(dF_)))}))
