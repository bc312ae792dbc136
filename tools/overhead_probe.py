import sys, time
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
Th = np.asfortranarray(np.tile(theta[:, None], (1, 64)))
for Ns in (2, 10000):
    for _ in range(5):
        vbmc_amd.negelcbo_batch(Th, 0, vp, gp, Ns, True, 0, seed=1, engine=eng)
    t = time.perf_counter()
    n = 50
    for i in range(n):
        vbmc_amd.negelcbo_batch(Th, 0, vp, gp, Ns, True, 0, seed=i, engine=eng)
    print("Ns", Ns, "ms/call", 1e3 * (time.perf_counter() - t) / n)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(200):
    vbmc_amd.negelcbo_batch(Th, 0, vp, gp, 2, True, 0, seed=i, engine=eng)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
