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


def run(R, mode):
    th = np.asfortranarray(theta0[:, None] + 0.05 * np.random.default_rng(100).standard_normal((theta0.size, R)))
    kw = dict(engine=eng, outputs=("F", "dF"))
    if mode != "rng":
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        shape = (1 if mode == "shared" else R, K, Ns // 2, D)
        eps_d = torch.randn(shape, dtype=torch.float64, device=dev, generator=g)
        if mode == "zeros":
            eps_d.zero_()
        torch.cuda.synchronize()
        kw.update(eps_device_ptr=eps_d.data_ptr(), eps_shared=(mode == "shared"))
    for _ in range(2):
        vbmc_amd.negelcbo_batch(th, 0, vp, gp, Ns, True, 0, **kw)
    eng.ctx.set_profiling(True)
    ems = []
    for _ in range(6):
        vbmc_amd.negelcbo_batch(th, 0, vp, gp, Ns, True, 0, **kw)
        ems.append(eng.ctx.last_kernel_ms()[0])
    eng.ctx.set_profiling(False)
    return float(np.median(ems))


for R in (64, 8):
    for mode in ("rng", "distinct", "shared"):
        print("R=%d %-9s entropy kernel %.3f ms" % (R, mode, run(R, mode)), flush=True)
