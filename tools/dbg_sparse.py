import sys, time
import numpy as np
sys.path.insert(0, '.')
import vbmc_amd
from bench import synth_inputs
D, N, K, Ns, S, Rr = 10, 400, 50, 10000, 20, 64
inp = synth_inputs(0, D, N, K, S)
eng = vbmc_amd.Engine(0)
gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng)
vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
theta0 = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
thetas = np.asfortranarray(theta0[:, None] + 0.05 * np.random.default_rng(1).standard_normal((theta0.size, Rr)))
mu = inp["mu"] / inp["lam"][:, None]
d = np.sqrt(((mu[:, :, None] - mu[:, None, :]) ** 2).sum(0))
np.fill_diagonal(d, np.inf)
print("sigma", inp["sigma"].min(), inp["sigma"].max(), "nn dist", d.min(1).min(), np.median(d.min(1)), "lam", inp["lam"])
eng.ctx.set_profiling(True)
for cut in (0.0, 100.0, 1e-3):
    for i in range(2):
        vbmc_amd.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, seed=5, engine=eng, sparse_cutoff=cut)
    o = vbmc_amd.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, seed=5, engine=eng, sparse_cutoff=cut)
    print("cutoff", cut, "ent_ms", eng.ctx.last_kernel_ms()[0], "H0", o["H"][0])
