"""Entropy-kernel duration with the component tail limited to 0 / 1 / 2 values per lane (VBMC_ENT_TAIL) over the shapes it
applies to (GPU box).   python tools/tail_sweep.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(D, K) for K in (18, 22, 24, 34, 38, 40, 50, 53, 56, 68, 76, 80, 100, 108, 112, 136, 150, 200, 216) for D in (6, 10, 20)]


def one():
    sys.path.insert(0, ROOT)
    import numpy as np

    import vbmc_amd
    eng = vbmc_amd.Engine(0)
    out = {}
    for D, K in SHAPES:
        rng = np.random.default_rng(D * 1000 + K)
        R, Ns = (32 if K <= 32 else 16 if K <= 64 else 8), 8192
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


def main():
    res = {}
    for t in ("0", "1", "2"):
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, VBMC_ENT_TAIL=t), capture_output=True, text=True)
        line = [ln for ln in o.stdout.splitlines() if ln.startswith("{")]
        res[t] = json.loads(line[-1]) if line else {}
        if not line:
            print(o.stderr[-400:])
    print("shape(D,K)   tail<=0   tail<=1   tail<=2   (kernel ms)")
    for D, K in SHAPES:
        k = "%d,%d" % (D, K)
        print("%2d,%3d  " % (D, K) + "  ".join("%8.3f" % res[t].get(k, float("nan")) for t in ("0", "1", "2")))


if __name__ == "__main__":
    one() if len(sys.argv) > 1 and sys.argv[1] == "one" else main()
