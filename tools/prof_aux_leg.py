"""rocprofv3 target: one leg of tools/bench_aux.py on its own, twelve calls.   python tools/prof_aux_leg.py sieve|pred|acqf|acqviqr|fullelcbo"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402

leg = sys.argv[1] if len(sys.argv) > 1 else "sieve"
D, N, K, S = 10, 400, 50, 20
inp = synth_inputs(0, D, N, K, S)
eng = vbmc_amd.Engine(0)
gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng)
vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
theta = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
Xs = 1.5 * np.random.default_rng(1).standard_normal((8192, D))
st = {"ymax": float(np.max(inp["y"])), "VarianceRegularizedAcqFcn": True, "TolGPVar": 1e-4}
legs = {
    "sieve": lambda: vbmc_amd.negelcbo_batch(np.tile(theta[:, None], (1, 250)), 0, vp, gp, 0, False, 0, engine=eng),
    "pred": lambda: vbmc_amd.gplite_pred(gp, Xs, None, None, True, engine=eng),
    "acqf": lambda: vbmc_amd.acqwrapper_vbmc(Xs, vp, gp, st, False, "acqf_vbmc", None, engine=eng),
    "fullelcbo": lambda: vbmc_amd.negelcbo_vbmc(theta, 0, vp, gp, 4096, 0, 1, nargout=11, engine=eng),
}
f = legs[leg]
for _ in range(4):
    f()
t = time.perf_counter()
for _ in range(12):
    f()
print(leg, "ms", 1e3 * (time.perf_counter() - t) / 12)
