"""The multi-GPU step's own cost on ONE device (GPU box): the pipelined step through a one-rank communicator (a real ncclAllGather,
k_comm_pick, the gathered read-back) beside the plain pipelined step of the same R on the same device, at the headline shape.
usage: python tools/r4_comm_probe.py [NAME=ENV1=v,ENV2=v ...]"""
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
    from vbmc_amd.multi import Comm

    D, N, K, S, Ns = 10, 400, 50, 20, 10000
    inp = synth_inputs(0, D, N, K, S)
    comm1 = Comm.create_all(1)
    eng = comm1.engines[0] if hasattr(comm1, "engines") else vbmc_amd.Engine(0)
    gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng)
    gps1 = comm1.upload_gp(gp)
    vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
    vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
    theta0 = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
    T = theta0.size
    DM = 1 if os.environ.get("VBMC_SLOT_STREAMS") == "0" else 3
    res = {}
    for R in [int(x) for x in os.environ.get("PROBE_R", "64,8").split(",")]:
        th = np.asfortranarray(theta0[:, None] + 0.05 * np.random.default_rng(100).standard_normal((T, R)))
        po = comm1.prepare(T, R, 0, vp, gps1, Ns)
        obj = vbmc_amd.PreparedObjective(T, R, 0, vp, gp, Ns, 0, None, engine=eng)

        def run_comm(n, i0):
            pend = []
            for i in range(n):
                po.submit(th, seed=i0 + i, slot=i & DM)
                pend.append(i & DM)
                if len(pend) == DM + 1:
                    F_, _ = po.collect(pend.pop(0))
                    np.argsort(F_, kind="stable")
            while pend:
                F_, _ = po.collect(pend.pop(0))
                np.argsort(F_, kind="stable")

        def run_plain(n, i0):
            for _ in obj.stream([th] * n, seeds=list(range(i0, i0 + n))):
                pass

        Fp, _ = obj(th, seed=5)
        po.submit(th, seed=5, slot=0)
        Fc, _ = po.collect(0)
        res["maxdiff%d" % R] = float(np.max(np.abs(Fp - Fc)))
        nst = 20 if R >= 32 else 60
        run_plain(4, 1)
        run_comm(4, 1)
        ts = {"plain": [], "comm": []}
        only = os.environ.get("PROBE_ONLY")
        for rep in range(int(os.environ.get("PROBE_REPS", "9"))):       # interleaved: the host's speed drifts between and within boxes
            for name, fn in (("plain", run_plain), ("comm", run_comm)):
                if only and name != only:
                    ts[name].append(1.0)
                    continue
                t1 = time.perf_counter()
                fn(nst, 100 * (rep + 1))
                ts[name].append((time.perf_counter() - t1) / nst)
        for name in ts:
            res["%s%d_ms" % (name, R)] = round(1e3 * float(np.median(ts[name])), 4)
            res["%s%d_min_ms" % (name, R)] = round(1e3 * float(np.min(ts[name])), 4)
        res["overhead%d_pct" % R] = round(100.0 * (res["comm%d_ms" % R] / res["plain%d_ms" % R] - 1.0), 2)
        res["overhead%d_min_pct" % R] = round(100.0 * (res["comm%d_min_ms" % R] / res["plain%d_min_ms" % R] - 1.0), 2)
    comm1.free_gp(gps1)
    comm1.close()
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
