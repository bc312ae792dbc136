"""A/B of the pipelined step at the headline shape for R = 64, 8 under environment settings (GPU box).
usage: python tools/r4_step_ab.py NAME=ENV1=v,ENV2=v ..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one():
    sys.path.insert(0, ROOT)
    import time

    import numpy as np

    import vbmc_amd
    from bench import synth_inputs

    D, N, K, S, Ns = 10, 400, 50, 20, int(os.environ.get("AB_NS", "10000"))
    inp = synth_inputs(0, D, N, K, S)
    eng = vbmc_amd.Engine(0)
    gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng)
    vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
    vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
    theta0 = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
    res = {}
    for R in [int(x) for x in os.environ.get("AB_R", "64,8").split(",")]:
        th = np.asfortranarray(theta0[:, None] + 0.05 * np.random.default_rng(100).standard_normal((theta0.size, R)))
        obj = vbmc_amd.PreparedObjective(theta0.size, R, 0, vp, gp, Ns, 0, None, engine=eng)
        F, dF = obj(th, seed=3)
        res["chk%d" % R] = float(F.sum()) + float(np.abs(dF).sum())
        for _ in obj.stream([th] * 4, seeds=[1, 2, 3, 4]):
            pass
        n = 20 if R >= 32 else 60
        ts = []
        for rep in range(5):
            t1 = time.perf_counter()
            for _ in obj.stream([th] * n, seeds=list(range(10, 10 + n))):
                pass
            ts.append((time.perf_counter() - t1) / n)
        res["ms%d" % R] = 1e3 * float(np.median(ts))
        res["evals_per_s%d" % R] = R / (res["ms%d" % R] * 1e-3)
    print(json.dumps(res))


def main():
    for spec in sys.argv[1:] or ["base="]:
        name, _, envs = spec.partition("=")
        env = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, **env), capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        print(name, json.loads(line[-1]) if line else r.stderr[-400:])


if __name__ == "__main__":
    one() if len(sys.argv) > 1 and sys.argv[1] == "one" else main()
