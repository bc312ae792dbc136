"""On-device Adam (vbmc_adam_batch) at VBMC's own sample count: us per iteration for R chains under environment settings (GPU box).
usage: python tools/r4_adam_probe.py NAME=ENV=v,ENV=v ..."""
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

    D, N, K, S = 10, 400, 50, 20
    inp = synth_inputs(0, D, N, K, S)
    eng = vbmc_amd.Engine(0)
    gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng)
    vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
    vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
    theta0 = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
    ns_v = int(np.ceil(100 * K ** (2.0 / 3.0) / K))
    vpb, tb = vbmc_amd.vpbounds(vp, {"X": inp["X"], "y": inp["y"]}, {"TolConLoss": 0.01, "TolWeight": 1e-2, "WeightPenalty": 0.1, "TolLength": 1e-6}, K)
    res = {}
    for R in (1, 2):
        x0 = np.asfortranarray(theta0[:, None] + 0.05 * np.random.default_rng(100).standard_normal((theta0.size, R)))
        vbmc_amd.fminadam_device(x0, 0, vpb, gp, ns_v, tb, 0.0, 60, seed=5, engine=eng)
        ts = []
        for rep in range(5):
            t1 = time.perf_counter()
            _, _, _, _, its = vbmc_amd.fminadam_device(x0, 0, vpb, gp, ns_v, tb, 0.0, 400, seed=6 + rep, engine=eng)
            ts.append((time.perf_counter() - t1) / float(np.max(its)))
        res["us_per_iteration_R%d" % R] = round(1e6 * float(np.median(ts)), 2)
    print(json.dumps(res))


def main():
    for spec in sys.argv[1:] or ["base="]:
        name, _, envs = spec.partition("=")
        env = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, **env), capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        print(name, json.loads(line[-1]) if line else r.stderr[-800:])


if __name__ == "__main__":
    one() if len(sys.argv) > 1 and sys.argv[1] == "one" else main()
