"""rocprofv3 target: gplite_nlZ + gradient at the C3 GP shape (N=400, D=10), B = 1 and B = 64."""
import sys

import numpy as np

sys.path.insert(0, ".")
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402

D, N, K, S = 10, 400, 50, 20
inp = synth_inputs(0, D, N, K, S)
eng = vbmc_amd.Engine(0)
gpd = {"X": inp["X"], "y": inp["y"], "s2": None, "covfun": 1, "Ncov": D + 1, "noisefun": (1, 0, 0), "Nnoise": 1, "meanfun": 4,
       "Nmean": 2 * D + 1, "intmeanfun": 0}
for B in (1, 64, 256):
    H = np.tile(inp["hyp"], (1, (B + S - 1) // S))[:, :B]
    for _ in range(10):
        vbmc_amd.gplite_nlZ(H, gpd, engine=eng)
