"""rocprofv3 target: eval_fullelcbo (misc/vpoptimize_vbmc.m:288-289: negelcbo with NSentFine samples, full variance, separate_K) at the
C3 shape, 30 calls."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402

D, N, K, S = 10, 400, 50, 20
inp = synth_inputs(0, D, N, K, S)
eng = vbmc_amd.Engine(0)
gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng)
vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
theta = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
for i in range(3):
    vbmc_amd.negelcbo_vbmc(theta, 0, vp, gp, 4096, 0, 1, nargout=11, engine=eng, seed=i)
t = time.perf_counter()
for i in range(30):
    vbmc_amd.negelcbo_vbmc(theta, 0, vp, gp, 4096, 0, 1, nargout=11, engine=eng, seed=10 + i)
print("ms/call", 1e3 * (time.perf_counter() - t) / 30)
