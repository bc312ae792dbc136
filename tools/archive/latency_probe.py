"""Wall time of the GP-side calls at small batch sizes (the acquisition optimiser and the rank-one update call them with
1 ... a few hundred points): python tools/latency_probe.py"""
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
rng = np.random.default_rng(0)


def timeit(f, n=20):
    f(); f()
    t = time.perf_counter()
    for _ in range(n):
        f()
    return 1e3 * (time.perf_counter() - t) / n


out = {}
optimState = {"ymax": float(np.max(inp["y"])), "VarianceRegularizedAcqFcn": True, "TolGPVar": 1e-4, "gpLengthscale": 1.0}
for ns in (1, 16, 128, 1024, 8192):
    Xs = 1.5 * rng.standard_normal((ns, D))
    out["pred_%d_ms" % ns] = timeit(lambda: vbmc_amd.gplite_pred(gp, Xs, None, None, False, engine=eng))
    try:
        out["acqf_%d_ms" % ns] = timeit(lambda: vbmc_amd.acqwrapper_vbmc(Xs, vp, gp, optimState, False, "acqf_vbmc", engine=eng))
    except Exception as e:  # noqa: BLE001
        out["acqf_err"] = repr(e)[:200]
x1 = 1.5 * rng.standard_normal(D)
out["rank1_update_ms"] = timeit(lambda: vbmc_amd.gplite_post_rank1(gp, x1, 0.3, engine=eng), 5)
out["rank1_update_device_only_ms"] = timeit(lambda: vbmc_amd.gplite_post_rank1(gp, x1, 0.3, need_L=False, engine=eng), 5)
g = gp
t = time.perf_counter()
for i in range(10):   # the active-sampling pattern: each append starts from the previous one
    g = vbmc_amd.gplite_post_rank1(g, 1.5 * rng.standard_normal(D), 0.1 * i, need_L=False, engine=eng)
out["rank1_chain_device_only_ms"] = 1e2 * (time.perf_counter() - t)
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in out.items()})
