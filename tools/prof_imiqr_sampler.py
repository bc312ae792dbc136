"""Wall time of the importance sampler behind acqimiqr_vbmc (private/activeimportancesampling_vbmc.m:103-246) at the headline GP
shape with VBMC's default options (100 + 100 resampling points, 100 MCMC samples per hyper-sample, W = 2 (D + 1) walkers):
all S ensembles advance in lock-step, every log-density evaluation is one batched device prediction.   python tools/prof_imiqr_sampler.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402

D, N, K, S = 10, 400, 50, 20
inp = synth_inputs(0, D, N, K, S, noisy=True)
eng = vbmc_amd.Engine(0)
gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, inp.get("noisefun", (1, 1, 0)), inp.get("s2"), engine=eng)
vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
for rep in range(3):
    t = time.perf_counter()
    ais = vbmc_amd.activeimportancesampling_vbmc(vp, gp, "acqimiqr_vbmc", None, {}, rng=np.random.default_rng(rep), engine=eng)
    dt = time.perf_counter() - t
    print("activeimportancesampling_vbmc(acqimiqr): %.1f ms, %d target evaluations (each one point of one hyper-sample's chain), Xa %s"
          % (1e3 * dt, ais["funccount"], ais["Xa"].shape), flush=True)
