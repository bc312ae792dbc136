"""The blocking call (vbmc_elbo_batch: what a MATLAB call through the MEX gateway is) at the headline shape under environment settings
(GPU box): ms per call, and the spans of the entropy and log-joint kernels inside it.
usage: python tools/r4_blocking_probe.py NAME=ENV1=v,ENV2=v ..."""
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

    D, N, K, S, Ns = 10, 400, 50, 20, 10000
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
        if os.environ.get("AB_PIPE_FIRST") == "1":      # the slot streams exist before the first blocking call
            for _ in obj.stream([th] * 8, seeds=list(range(8))):
                pass
        for i in range(12):
            obj(th, seed=i)
        ts = []
        for i in range(30):
            t1 = time.perf_counter()
            obj(th, seed=100 + i)
            ts.append(time.perf_counter() - t1)
        res["call_ms_%d" % R] = round(1e3 * float(np.median(ts)), 4)
        eng.ctx.set_profiling(1)
        e, l = [], []
        for i in range(10):
            obj(th, seed=200 + i)
            a, b = eng.ctx.last_kernel_ms()
            e.append(a)
            l.append(b)
        eng.ctx.set_profiling(False)
        res["ent_span_%d" % R] = round(float(np.median(e)), 4)
        res["lj_span_%d" % R] = round(float(np.median(l)), 4)
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
