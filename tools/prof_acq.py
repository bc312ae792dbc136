"""rocprofv3 target: acquisition sweeps at the C3 GP shape (8192 points, S=20): acqf and acqviqr (Na=100)."""
import sys

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
Xs = 1.5 * np.random.default_rng(0).standard_normal((8192, D))
st = {"ymax": float(np.max(inp["y"])), "VarianceRegularizedAcqFcn": True, "TolGPVar": 1e-4}
gl = np.exp(np.mean(inp["hyp"][:D], axis=1))
gpn = dict(gp, X_rescaled=inp["X"] / gl[None, :], sn2new=np.full(N, 0.05))
stv = dict(st, gplengthscale=gl, ActiveImportanceSampling={"Xa": 1.2 * np.random.default_rng(2).standard_normal((100, D))})
for _ in range(5):
    vbmc_amd.acqwrapper_vbmc(Xs, vp, gp, st, False, "acqf_vbmc", None, engine=eng)
    vbmc_amd.acqwrapper_vbmc(Xs, vp, gpn, stv, False, "acqviqr_vbmc", None, engine=eng)
