"""rocprofv3 target: gplite_post with the factors left on the device (need_L=False) at N = 400, D = 10, S = 20, twelve calls;
then gplite_nlZ value + gradient for one hyper-parameter vector, twelve calls."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402

D, N, K, S = 10, 400, 50, 20
inp = synth_inputs(0, D, N, K, S)
eng = vbmc_amd.Engine(0)
for _ in range(4):
    vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, need_L=False, engine=eng)
t = time.perf_counter()
for _ in range(12):
    vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, need_L=False, engine=eng)
print("gplite_post resident ms", 1e3 * (time.perf_counter() - t) / 12)
gpd = {"X": inp["X"], "y": inp["y"], "s2": None, "covfun": 1, "Ncov": D + 1, "noisefun": (1, 0, 0), "Nnoise": 1, "meanfun": 4,
       "Nmean": 2 * D + 1, "intmeanfun": 0}
H = inp["hyp"][:, :1]
for _ in range(4):
    vbmc_amd.gplite_nlZ(H, gpd, engine=eng)
t = time.perf_counter()
for _ in range(12):
    vbmc_amd.gplite_nlZ(H, gpd, engine=eng)
print("nlZ+grad B=1 ms", 1e3 * (time.perf_counter() - t) / 12)
