"""Duration of the expected-log-joint kernel alone (no overlap with the entropy kernel) at small batches, for the kernel variants
(GPU box).  usage: python tools/r4_lj_probe.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one():
    sys.path.insert(0, ROOT)
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
    for R in (64, 16, 8, 4):
        th = np.asfortranarray(theta0[:, None] + 0.05 * np.random.default_rng(100).standard_normal((theta0.size, R)))
        obj = vbmc_amd.PreparedObjective(theta0.size, R, 0, vp, gp, Ns, 0, None, engine=eng)
        for i in range(3):
            obj(th, seed=i)
        eng.ctx.set_profiling(True)
        lj, ent = [], []
        for i in range(7):
            obj(th, seed=10 + i)
            e, l = eng.ctx.last_kernel_ms()
            lj.append(l); ent.append(e)
        eng.ctx.set_profiling(False)
        res["R%d" % R] = {"lj_us": 1e3 * float(np.median(lj)), "ent_us": 1e3 * float(np.median(ent))}
    print(json.dumps(res))


def main():
    for name, env in (("mfma", {"VBMC_LJ_OVERLAP": "0", "VBMC_LJ_KERNEL": "mfma"}), ("valu", {"VBMC_LJ_OVERLAP": "0", "VBMC_LJ_KERNEL": "valu"}),
                      ("default_no_overlap", {"VBMC_LJ_OVERLAP": "0"})):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, **env), capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        print(name, json.loads(line[-1]) if line else r.stderr[-400:])


if __name__ == "__main__":
    one() if len(sys.argv) > 1 and sys.argv[1] == "one" else main()
