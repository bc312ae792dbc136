"""gplite_pred call by call (the bimodal 1.0 / 3.8 ms of aux.gplite_pred_8192_ms): twenty consecutive calls on a fresh
engine, wall time of each, with and without the caller's result arrays reused; the pool's growth is what the first calls pay."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402

D, N, K, S = 10, 400, 50, 20
inp = synth_inputs(0, D, N, K, S)
for need_L in (True, False):
    eng = vbmc_amd.Engine(0)
    gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, need_L=need_L, engine=eng)
    Xs = 1.5 * np.random.default_rng(0).standard_normal((8192, D))
    ts = []
    for i in range(20):
        t = time.perf_counter()
        vbmc_amd.gplite_pred(gp, Xs, None, None, False, engine=eng)
        ts.append(1e3 * (time.perf_counter() - t))
    print("gplite_post(need_L=%s) then 20 x gplite_pred(8192 x 20), ms per call: %s" % (need_L, " ".join("%.2f" % x for x in ts)))
