"""One wave or two per workgroup for 32 < K <= 64?  Entropy-only evaluations (value + gradient, device RNG, Ns = 8192, R = 16),
kernel duration by HIP events, VBMC_ENT_HV=1 / 2 in separate processes.   python tools/hv_small_sweep.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(D, K) for D in (6, 10, 14, 16, 18, 20, 24, 28, 32) for K in (36, 40, 48, 52, 56, 64)]


def one():
    sys.path.insert(0, ROOT)
    import numpy as np

    import vbmc_amd
    eng = vbmc_amd.Engine(0)
    out = {}
    for D, K in SHAPES:
        rng = np.random.default_rng(D * 1000 + K)
        R, Ns = 16, 8192
        mu = 1.5 * rng.standard_normal((D, K))
        vp = vbmc_amd.make_vp(mu, 0.3 * np.exp(0.2 * rng.standard_normal(K)), np.ones(D), eta=0.3 * rng.standard_normal(K))
        vp["w"] = np.exp(vp["eta"]) / np.sum(np.exp(vp["eta"]))
        theta = np.concatenate([mu.reshape(-1, order="F"), np.log(vp["sigma"]).reshape(-1), np.log(vp["lambda"]).reshape(-1), vp["eta"].reshape(-1)])
        th = np.asfortranarray(theta[:, None] + 0.02 * rng.standard_normal((theta.size, R)))
        for i in range(2):
            vbmc_amd.negelcbo_batch(th, 0, vp, None, Ns, True, 0, seed=i, engine=eng, outputs=("H",))
        eng.ctx.set_profiling(True)
        ms = []
        for i in range(5):
            vbmc_amd.negelcbo_batch(th, 0, vp, None, Ns, True, 0, seed=10 + i, engine=eng, outputs=("H",))
            ms.append(eng.ctx.last_kernel_ms()[0])
        eng.ctx.set_profiling(False)
        out["%d,%d" % (D, K)] = float(np.median(ms))
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        res = {}
        for hv in ("1", "2"):
            r = subprocess.run([sys.executable, __file__, "one"], env=dict(os.environ, VBMC_ENT_HV=hv), capture_output=True, text=True)
            res[hv] = json.loads(r.stdout.strip().splitlines()[-1])
        print("| D | K | one wave ms | two waves ms | two / one |")
        print("|---|---|---|---|---|")
        for D, K in SHAPES:
            a, b = res["1"]["%d,%d" % (D, K)], res["2"]["%d,%d" % (D, K)]
            print("| %d | %d | %.3f | %.3f | %.2f |" % (D, K, a, b, b / a))
