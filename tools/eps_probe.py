"""Why is the entropy kernel slower when it reads the caller's draws (eps_mode 1 / 2) than when it draws on the device?
Headline shape; kernel time by HIP events (ctx profiling) for: device RNG; R distinct blocks in HBM (1.28 GB); ONE shared block
(20 MB: L2 / MALL resident); R distinct blocks with R = 8 (160 MB).   python tools/eps_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402

D, N, K, Ns, S = 10, 400, 50, 10000, 20
inp = synth_inputs(0, D, N, K, S)
eng = vbmc_amd.Engine(0)
gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng)
vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
theta0 = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
dev = torch.device("cuda:0")


def setup(R, mode):
    th = np.asfortranarray(theta0[:, None] + 0.05 * np.random.default_rng(100).standard_normal((theta0.size, R)))
    kw = dict(engine=eng, outputs=("F", "dF"))
    keep = None
    if mode != "rng":
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        shape = (1 if mode == "shared" else R, K, Ns // 2, D)
        keep = torch.randn(shape, dtype=torch.float64, device=dev, generator=g)
        torch.cuda.synchronize()
        kw.update(eps_device_ptr=keep.data_ptr(), eps_shared=(mode == "shared"))
    return th, kw, keep


def measure(th, kw, n=6):
    eng.ctx.set_profiling(2)
    ems = []
    for _ in range(n):
        vbmc_amd.negelcbo_batch(th, 0, vp, gp, Ns, True, 0, **kw)
        ems.append(eng.ctx.last_kernel_ms()[0])
    eng.ctx.set_profiling(False)
    return ems


for R in (64, 8):
    cases = {m: setup(R, m) for m in ("rng", "distinct", "shared")}
    for m in cases:                       # warm: code pages, clocks
        measure(*cases[m][:2], n=12)
    ems = {m: [] for m in cases}
    for rnd in range(4):                  # interleaved: the clocks drift over a run
        for m in cases:
            ems[m] += measure(*cases[m][:2])
    for m in cases:
        print("R=%d %-9s entropy kernel median %.3f min %.3f ms" % (R, m, float(np.median(ems[m])), float(np.min(ems[m]))), flush=True)
    del cases
