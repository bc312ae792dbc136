"""Timing of the non-headline device paths at the C3 shape: gplite_post, gplite_pred (2^13 points), full-variance ELCBO."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402

D, N, K, S = 10, 400, 50, 20
inp = synth_inputs(0, D, N, K, S)
eng = vbmc_amd.Engine(0)


def timeit(f, n=5):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n


out = {}
out["gplite_post_ms"] = 1e3 * timeit(lambda: vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng), 3)
gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng)
Xs = 1.5 * np.random.default_rng(0).standard_normal((8192, D))
out["gplite_pred_8192_ms"] = 1e3 * timeit(lambda: vbmc_amd.gplite_pred(gp, Xs, None, None, False, engine=eng), 3)
vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
theta = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
out["eval_fullelcbo_ms"] = 1e3 * timeit(lambda: vbmc_amd.negelcbo_vbmc(theta, 0, vp, gp, 4096, 0, 1, nargout=11, engine=eng), 5)
out["diagvar_grad_ms"] = 1e3 * timeit(lambda: vbmc_amd.negelcbo_vbmc(theta, 1.0, vp, gp, 128, 1, 2, nargout=2, engine=eng), 5)
out["entlb_sieve_R250_ms"] = 1e3 * timeit(lambda: vbmc_amd.negelcbo_batch(np.tile(theta[:, None], (1, 250)), 0, vp, gp, 0, False, 0, engine=eng), 5)
print(json.dumps(out))
