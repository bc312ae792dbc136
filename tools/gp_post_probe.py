#!/usr/bin/env python
"""Where does one gplite_post call spend its wall time?  20 calls with the factors left on the device (need_L=False) at the C3 GP
shape; run under `rocprofv3 --kernel-trace --hip-trace --stats` for the per-API / per-kernel split."""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (first: see README)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402

torch.cuda.init()
inp = synth_inputs(0, 10, 400, 50, 20)
eng = vbmc_amd.Engine(0)
need_L = bool(int(os.environ.get("NEED_L", "0")))
n = int(os.environ.get("CALLS", "20"))
f = lambda: vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, need_L=need_L, engine=eng)  # noqa: E731
for _ in range(6):
    f()
ts = []
for _ in range(n):
    t0 = time.perf_counter()
    f()
    ts.append(1e3 * (time.perf_counter() - t0))
print("gplite_post need_L=%d: median %.3f ms, min %.3f, max %.3f" % (need_L, float(np.median(ts)), min(ts), max(ts)))
