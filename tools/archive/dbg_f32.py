import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import vbmc_amd
from vbmc_amd import _lib
from tests._cases import synth_problem
print("lib", _lib.LIB_PATH)
D, N, K, S, Ns = 20, 100, 100, 2, 2000
p = synth_problem(41, D, N, K, S, noisy=True)
gp = vbmc_amd.gplite_post(p["hyp"], p["X"], p["y"], 1, 4, p["noisefun"], p["s2"])
vp = vbmc_amd.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"]); vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
theta = np.concatenate([p["mu"].reshape(-1, order="F"), np.log(p["sigma"]), np.log(p["lam"]), p["eta"]])
a = vbmc_amd.negelcbo_batch(theta, 0, vp, gp, Ns, True, 0, seed=2024)
print("H %.17g dH0 %.17g" % (a["H"][0], a["dH"][0, 0]))
b = vbmc_amd.negelcbo_batch(theta, 0, vp, None, Ns, True, 0, seed=2024)
print("Hent %.17g" % b["H"][0])
