"""rocprofv3 target: one Adam chain entirely on the device (vbmc_adam_batch, R = 1) at the C3 shape, 200 iterations."""
import sys

import numpy as np

sys.path.insert(0, ".")
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402

import os
D, N, K, S, Ns = int(os.environ.get("PROF_D", "10")), 400, int(os.environ.get("PROF_K", "50")), 20, int(os.environ.get("PROF_NS", "10000"))
inp = synth_inputs(0, D, N, K, S)
eng = vbmc_amd.Engine(0)
gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng)
vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
theta = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
x0 = np.asfortranarray(theta[:, None])
# soft bounds as vpoptimize_vbmc passes them (misc/vpbounds.m): the production call has thetabnd
opts = {"TolConLoss": 0.01, "TolWeight": 1e-2, "WeightPenalty": 0.1, "TolLength": 1e-6}
try:
    vp, thetabnd = vbmc_amd.vpbounds(vp, {"X": inp["X"], "y": inp["y"]}, opts, K)
except Exception as e:  # noqa: BLE001
    print("no bounds:", e)
    thetabnd = None
vbmc_amd.fminadam_device(x0, 0, vp, gp, Ns, thetabnd, 0.0, 40, seed=5, engine=eng)
import time
t = time.perf_counter()
_, _, _, _, its = vbmc_amd.fminadam_device(x0, 0, vp, gp, Ns, thetabnd, 0.0, 200, seed=6, engine=eng)
print("us/iter", 1e6 * (time.perf_counter() - t) / int(its[0]))
