"""rocprofv3 target: the ELBO with the diagonal BQ variance and its gradient (negelcbo_vbmc(theta, beta = 1, ..., compute_var = 2,
gradient) at the headline GP shape, Ns = 128), twelve calls -- tools/bench_aux.py's diagvar_grad leg on its own."""
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
f = lambda: vbmc_amd.negelcbo_vbmc(theta, 1.0, vp, gp, 128, 1, 2, nargout=2, engine=eng)  # noqa: E731
for _ in range(4):
    f()
t = time.perf_counter()
for _ in range(12):
    f()
print("negelcbo + diagonal variance + gradient ms", 1e3 * (time.perf_counter() - t) / 12)
