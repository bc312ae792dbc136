"""Does the pipelined step depend on which hardware queues the process created before the slot streams?  (GPU box)
AB_DUMMY = k: k torch streams are created (and used once) between the context and its first pipelined submit; VBMC_PLACE=0 turns the
measured placement off.   usage: python tools/r4_place_check.py NAME=ENV=v,ENV=v ..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one():
    sys.path.insert(0, ROOT)
    import time

    import numpy as np
    import torch

    torch.cuda.init()
    import vbmc_amd
    from bench import synth_inputs

    D, N, K, S, Ns = 10, 400, 50, 20, 10000
    inp = synth_inputs(0, D, N, K, S)
    eng = vbmc_amd.Engine(0)
    gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng)
    vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
    vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
    theta0 = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
    keep = []
    for _ in range(int(os.environ.get("AB_DUMMY", "0"))):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            keep.append(torch.zeros(8, device="cuda") + 1)
        s.synchronize()
        keep.append(s)
    res = {}
    for R in (64, 8):
        th = np.asfortranarray(theta0[:, None] + 0.05 * np.random.default_rng(100).standard_normal((theta0.size, R)))
        obj = vbmc_amd.PreparedObjective(theta0.size, R, 0, vp, gp, Ns, 0, None, engine=eng)
        for _ in obj.stream([th] * 8, seeds=list(range(8))):
            pass
        n = 20 if R >= 32 else 60
        ts = []
        for rep in range(5):
            t1 = time.perf_counter()
            for _ in obj.stream([th] * n, seeds=list(range(10, 10 + n))):
                pass
            ts.append((time.perf_counter() - t1) / n)
        res["ms%d" % R] = round(1e3 * float(np.median(ts)), 4)
    print(json.dumps(res))


def main():
    for spec in sys.argv[1:] or ["base="]:
        name, _, envs = spec.partition("=")
        env = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, **env), capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        place = [ln for ln in r.stderr.splitlines() if ln.startswith("[vbmc place]")]
        print(name, json.loads(line[-1]) if line else r.stderr[-800:], " ".join(p.split(": ")[-1] for p in place))


if __name__ == "__main__":
    one() if len(sys.argv) > 1 and sys.argv[1] == "one" else main()
