"""Host-side cost of ONE objective call (the MATLAB fminadam pattern: one negelcbo_vbmc per iteration, R = 1) at the C3
shape: the prepared objective with the full MC sample count, and with Ns = 2 (device work ~ 0: what is left is the
launch / copy / synchronise floor of the call)."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np

import vbmc_amd
from bench import synth_inputs

D, N, K, S = 10, 400, 50, 20
inp = synth_inputs(0, D, N, K, S)
eng = vbmc_amd.Engine(0)
gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng)
vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
theta = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
out = {}
for Ns in (2, 10000):
    obj = vbmc_amd.PreparedObjective(theta.size, 1, 0.0, vp, gp, Ns, 0, None, engine=eng)
    for i in range(20):
        obj(theta, seed=i)
    n = 500
    t = time.perf_counter()
    for i in range(n):
        obj(theta, seed=i)
    out["prepared_R1_Ns%d_us" % Ns] = 1e6 * (time.perf_counter() - t) / n
print(out)
