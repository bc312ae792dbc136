"""gplite_pred on the GPU box: wall time per call and per-kernel time (HIP events are not exposed here: wall time of warmed calls),
fused form against the two-kernel form (VBMC_PRED_FUSED=0), with the difference of their outputs.  Usage: python tools/pred_probe.py [D N S Nstar]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402

D, N, S, Nstar = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (10, 400, 20, 8192)))
eng = vbmc_amd.default_engine()
inp = synth_inputs(0, D, N, 50, S)
gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, need_L=False, engine=eng)
Xs = 1.5 * np.random.default_rng(0).standard_normal((Nstar, D))
out = vbmc_amd.gplite_pred(gp, Xs, None, None, False, engine=eng)
for _ in range(5):
    vbmc_amd.gplite_pred(gp, Xs, None, None, False, engine=eng)
ts = []
for _ in range(15):
    t1 = time.perf_counter()
    o = vbmc_amd.gplite_pred(gp, Xs, None, None, False, engine=eng)
    ts.append(time.perf_counter() - t1)
print(json.dumps({"env": os.environ.get("VBMC_PRED_FUSED", "1"), "shape": [D, N, S, Nstar], "ms_median_min": [1e3 * float(np.median(ts)), 1e3 * float(np.min(ts))],
                  "sum_fmu": float(np.sum(o[2])), "sum_fs2": float(np.sum(o[3]))}))
np.save("/tmp/pred_%s.npy" % os.environ.get("VBMC_PRED_FUSED", "1"), np.stack([np.asarray(o[2]).reshape(-1), np.asarray(o[3]).reshape(-1)]))
if os.path.exists("/tmp/pred_0.npy") and os.path.exists("/tmp/pred_1.npy"):
    a, b = np.load("/tmp/pred_0.npy"), np.load("/tmp/pred_1.npy")
    if a.shape == b.shape:
        print("max |fmu diff| %.3e (scale %.3e)  max |fs2 diff| %.3e (scale %.3e)" % (np.max(np.abs(a[0] - b[0])), np.max(np.abs(a[0])), np.max(np.abs(a[1] - b[1])), np.max(np.abs(a[1]))))
