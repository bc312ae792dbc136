"""Entropy-kernel duration over a grid of mixture shapes for each tuning variant built by tools/tune_build.py (GPU box):

    python tools/tune_sweep.py [small]    -> table: shape x variant, kernel ms (entropy-only evaluations, R restarts batched);
                                             "small": K <= 64 only, and no forced four-wave column

Each variant runs in its own process (the library is chosen at import through VBMC_HIP_LIB)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = "small" in sys.argv[1:] or os.environ.get("TUNE_SMALL") == "1"
SHAPES = ([(D, K) for K in (8, 16, 24, 32, 40, 48, 56, 64) for D in (2, 6, 10, 14, 20, 28)] if SMALL else
          [(D, K) for K in (48, 64, 80, 96, 112, 128, 192, 256) for D in (6, 10, 14, 18, 20, 24, 28, 32)])
if os.environ.get("TUNE_KS"):      # another grid: TUNE_KS=20,36,52 TUNE_DS=6,10,14
    SHAPES = [(D, K) for K in map(int, os.environ["TUNE_KS"].split(",")) for D in map(int, os.environ.get("TUNE_DS", "6,10,14,18,20,24,28,32").split(","))]


def one():
    sys.path.insert(0, ROOT)
    import numpy as np

    import vbmc_amd
    eng = vbmc_amd.Engine(0)
    out = {}
    for D, K in SHAPES:
        rng = np.random.default_rng(D * 1000 + K)
        R, Ns = (64 if K <= 16 else 16 if K <= 64 else 8), 8192     # steady state: >= 30 tiles per wave after chunking
        mu = 1.5 * rng.standard_normal((D, K))
        vp = vbmc_amd.make_vp(mu, 0.3 * np.exp(0.2 * rng.standard_normal(K)), np.ones(D), eta=0.3 * rng.standard_normal(K))
        vp["w"] = np.exp(vp["eta"]) / np.sum(np.exp(vp["eta"]))
        theta = np.concatenate([mu.reshape(-1, order="F"), np.log(vp["sigma"]).reshape(-1), np.log(vp["lambda"]).reshape(-1), vp["eta"].reshape(-1)])
        th = np.asfortranarray(theta[:, None] + 0.02 * rng.standard_normal((theta.size, R)))
        try:
            for i in range(2):
                vbmc_amd.negelcbo_batch(th, 0, vp, None, Ns, True, 0, seed=i, engine=eng, outputs=("H",))
            eng.ctx.set_profiling(True)
            ms = []
            for i in range(5):
                vbmc_amd.negelcbo_batch(th, 0, vp, None, Ns, True, 0, seed=10 + i, engine=eng, outputs=("H",))
                ms.append(eng.ctx.last_kernel_ms()[0])
            eng.ctx.set_profiling(False)
            out["%d,%d" % (D, K)] = [float(np.median(ms)), Ns]
        except Exception as e:  # noqa: BLE001
            out["%d,%d" % (D, K)] = [None, str(e)[:60]]
    print(json.dumps(out))


def main():
    tune = os.path.join(ROOT, "vbmc_amd", "lib", "tune")
    libs = sorted(f for f in os.listdir(tune) if f.endswith(".so"))
    res = {}
    for lib in libs:
        for hv in (("",) if (SMALL or os.environ.get("TUNE_NO_HV")) else ("", "2", "4")):
            env = dict(os.environ, VBMC_HIP_LIB=os.path.join(tune, lib), TUNE_SMALL="1" if SMALL else "0")
            if hv:
                env["VBMC_ENT_HV"] = hv
            o = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, capture_output=True, text=True)
            line = [ln for ln in o.stdout.splitlines() if ln.startswith("{")]
            res[lib[4:-3] + ("/hv" + hv if hv else "")] = json.loads(line[-1]) if line else {"error": o.stderr[-300:]}
    names = list(res)
    print("shape(D,K)  Ns   " + "  ".join("%8s" % n for n in names))
    for D, K in SHAPES:
        key = "%d,%d" % (D, K)
        row = [res[n].get(key, [None])[0] for n in names]
        ns = next((res[n][key][1] for n in names if key in res[n] and res[n][key][0] is not None), "")
        print("%2d,%3d %6s  " % (D, K, ns) + "  ".join("%8s" % ("%.3f" % v if v is not None else "-") for v in row))


if __name__ == "__main__":
    one() if len(sys.argv) > 1 and sys.argv[1] == "one" else main()
