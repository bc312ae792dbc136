"""rocprofv3 target: the rank-one append of an observation (gplite_post(gp, xstar, ystar, [], 1): gplite/gplite_post.m:173-251) on the
device surrogate at N = 400, D = 10, S = 20, twelve appends of one point each (N grows 400 -> 412)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402

D, N, K, S = 10, 400, 50, 20
inp = synth_inputs(0, D, N, K, S)
eng = vbmc_amd.Engine(0)
gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, need_L=False, engine=eng)
rng = np.random.default_rng(3)
for _ in range(3):
    gp = vbmc_amd.gplite_post_rank1(gp, 1.2 * rng.standard_normal((1, D)), float(rng.standard_normal()), need_L=False, engine=eng)
t = time.perf_counter()
for _ in range(12):
    gp = vbmc_amd.gplite_post_rank1(gp, 1.2 * rng.standard_normal((1, D)), float(rng.standard_normal()), need_L=False, engine=eng)
print("rank-one append ms", 1e3 * (time.perf_counter() - t) / 12)
