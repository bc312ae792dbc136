"""Host time of the pipelined step against its wall time (GPU box): seconds spent inside submit / collect per step and the step itself,
at shapes where the step is short.   python tools/r4_host_cost.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402

eng = vbmc_amd.Engine(0)
for name, (D, N, K, S, Ns, R) in {"headline R=1": (10, 400, 50, 20, 10000, 1), "headline R=2": (10, 400, 50, 20, 10000, 2),
                                   "headline R=8": (10, 400, 50, 20, 10000, 8), "configs[1] R=64": (6, 200, 10, 8, 1000, 64),
                                   "VBMC's own Ns, R=1": (10, 400, 50, 20, 28, 1)}.items():
    inp = synth_inputs(0, D, N, K, S)
    gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng)
    vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
    vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
    theta0 = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
    th = np.asfortranarray(theta0[:, None] + 0.05 * np.random.default_rng(100).standard_normal((theta0.size, R)))
    obj = vbmc_amd.PreparedObjective(theta0.size, R, 0, vp, gp, Ns, 0, None, engine=eng)
    for depth in (4,):
        def run(n, timed):
            ts = tc = 0.0
            pend = []
            for i in range(n):
                t = time.perf_counter()
                obj.submit(th, seed=i, slot=i % depth)
                ts += time.perf_counter() - t
                pend.append(i % depth)
                if len(pend) == depth:
                    t = time.perf_counter()
                    obj.collect(pend.pop(0))
                    tc += time.perf_counter() - t
            while pend:
                obj.collect(pend.pop(0))
            return ts, tc
        run(40, False)
        n = 200
        t0 = time.perf_counter()
        ts, tc = run(n, True)
        wall = time.perf_counter() - t0
        print("%-20s depth %d: step %.1f us   in submit %.1f us   in collect %.1f us (waiting included)" %
              (name, depth, 1e6 * wall / n, 1e6 * ts / n, 1e6 * tc / n), flush=True)
