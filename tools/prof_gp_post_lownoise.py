"""gplite_post on the low-noise branch (min(sn2) < 1e-6: L = -inv(K + sn2 I), gplite_core.m:84-99 -- the branch a deterministic target takes)
at N = 400, D = 10, S = 20, posterior left on the device; twelve calls."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402

D, N, K, S = 10, 400, 50, 20
inp = synth_inputs(0, D, N, K, S)
hyp = inp["hyp"].copy()
hyp[D + 1, :] = -8.0                      # ln sn: sn2 = e^-16 < 1e-6
eng = vbmc_amd.Engine(0)
f = lambda: vbmc_amd.gplite_post(hyp, inp["X"], inp["y"], 1, 4, (1, 0, 0), None, need_L=False, engine=eng)  # noqa: E731
for _ in range(4):
    gp = f()
print("Lchol flags", [p["Lchol"] for p in gp["post"]][:4], "sn2_mult", [p["sn2_mult"] for p in gp["post"]][:4])
t = time.perf_counter()
for _ in range(12):
    f()
print("gplite_post low-noise resident ms", 1e3 * (time.perf_counter() - t) / 12)
