"""A/B experiments on the headline entropy kernel (k_entropy_mfma<3,3,...>, D = 10, K = 50): library variants built with
extra -D flags on the QS = 3 translation unit, each timed under a list of environment settings, with a parity check against
the first variant on the same device stream (same seed: H and dH must agree to the summation-order level).  (Round 4: tools/r4_experiments.py.)

    python tools/ent_ab.py build            (here: cross-compiles vbmc_amd/lib/exp/libvbmc_hip_<name>.so)
    python tools/ent_ab.py run [R] [Ns]     (GPU box; ENT_AB_REPS=n: every variant n times, interleaved, median reported)

Variants: the table below, or ENT_AB="name:-Dflag,-Dflag;name2:...;head:@HEAD" (@HEAD: the entropy translation unit of the previous
round's last commit, unpacked under /tmp/head_src by `git archive <commit> vbmc_amd/csrc include | tar -x -C /tmp/head_src`).
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "vbmc_amd", "lib", "exp")
OBJ = os.path.join(ROOT, "vbmc_amd", "lib", "obj")
# name -> (compile flags, [environment settings to time it under])
VARIANTS = {
    "head": (["@HEAD"], [{}]),            # the committed kernel
    "tree": ([], [{}]),                   # the working tree
}      # (round 5's variants -- -DVBMC_NO_C2, -DVBMC_EXP_CUBIC, -DVBMC_NO_GP2, -DVBMC_EVX, -DVBMC_NO_ETZ, -DVBMC_EXP_TABLIN -- were switches of the
       #  product header until round 6; their results are profiles/r05_experiments.md, their code is in the history)
if os.environ.get("ENT_AB"):
    VARIANTS = {}
    for item in os.environ["ENT_AB"].split(";"):
        nm, _, fl = item.partition(":")
        VARIANTS[nm.strip()] = ([f for f in fl.split(",") if f], [{}])


def build(qs=3):
    only = os.environ.get("VBMC_EXP_ONLY")
    names = [k for k in VARIANTS if (not only or k in only.split(","))]
    os.makedirs(EXP, exist_ok=True)
    others = [os.path.join(OBJ, "vbmc_hip.o")] + [os.path.join(OBJ, "ent_mfma_qs%d.o" % q) for q in range(1, 10) if q != qs]
    others += [os.path.join(OBJ, "ent_lane_dt%d.o" % dt) for dt in (2, 4, 6, 8, 10, 12)]
    procs = []
    for name in names:
        flags = [f for f in VARIANTS[name][0] if f != "@HEAD"]
        src = os.path.join("/tmp/head_src" if "@HEAD" in VARIANTS[name][0] else ROOT, "vbmc_amd", "csrc", "ent_mfma_inst.hip")
        o = os.path.join(EXP, "ent_%s.o" % name)
        procs.append((name, o, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
                                                 "-Wno-pass-failed", "-DQS_VALUE=%d" % qs] + flags +
                                                ["-c", src, "-o", o])))
    for name, o, p in procs:
        assert p.wait() == 0, name
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", o] + others +
                              ["-ldl", "-o", os.path.join(EXP, "libvbmc_hip_%s.so" % name)])
        os.remove(o)
    print("built", names)


def one(R, Ns, dump):
    sys.path.insert(0, ROOT)
    import numpy as np

    if os.environ.get("EXP_EPS") == "1":
        import torch                       # before the library: the two HIP runtimes must come up in this order (as in bench.py)

        torch.cuda.init()

    import vbmc_amd
    from bench import synth_inputs

    D, N, K, S = int(os.environ.get("EXP_D", "10")), 400, int(os.environ.get("EXP_K", "50")), 20
    inp = synth_inputs(0, D, N, K, S)
    eng = vbmc_amd.Engine(0)
    gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng)
    vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
    vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
    theta0 = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
    th = np.asfortranarray(theta0[:, None] + 0.05 * np.random.default_rng(100).standard_normal((theta0.size, R)))
    out = None
    for i in range(3):
        out = vbmc_amd.negelcbo_batch(th, 0, vp, gp, Ns, True, 0, seed=7, engine=eng, outputs=("F", "dF", "H", "dH"))
    if dump:
        np.savez(dump, F=out["F"], dF=out["dF"], H=out["H"], dH=out["dH"])
    eng.ctx.set_profiling(True)
    ms = []
    if os.environ.get("ENT_AB_ALONE", "1") == "1":
        eng.ctx.set_profiling(2)       # nothing forked beside the kernel while it is timed (what bench.py's roofline leg prices)
    for i in range(int(os.environ.get("ENT_AB_ITERS", "20"))):
        vbmc_amd.negelcbo_batch(th, 0, vp, gp, Ns, True, 0, seed=10 + i, engine=eng, outputs=("F",))
        ms.append(eng.ctx.last_kernel_ms()[0])
    res = {"ms": float(np.median(ms)), "min": float(np.min(ms))}
    if os.environ.get("EXP_EPS") == "1":      # parity mode too: every restart its own block of draws, resident on the device
        import torch

        g = torch.Generator(device="cuda:0")
        g.manual_seed(1)
        eps_d = torch.randn((R, K, (Ns + 1) // 2, D), dtype=torch.float64, device="cuda:0", generator=g)
        torch.cuda.synchronize()
        kw = dict(eps_device_ptr=eps_d.data_ptr(), eps_shared=False, engine=eng, outputs=("F",))
        for _ in range(2):
            vbmc_amd.negelcbo_batch(th, 0, vp, gp, Ns, True, 0, **kw)
        ms2 = []
        for i in range(8):
            vbmc_amd.negelcbo_batch(th, 0, vp, gp, Ns, True, 0, **kw)
            ms2.append(eng.ctx.last_kernel_ms()[0])
        res["eps_ms"] = float(np.median(ms2))
    print(json.dumps(res))


def run(R, Ns):
    import numpy as np

    tmp = os.path.join(ROOT, "gpurun_out", "entab")
    os.makedirs(tmp, exist_ok=True)
    reps = int(os.environ.get("ENT_AB_REPS", "2"))
    tags, runs, notes = [], {}, {}
    ref = None
    for rep in range(reps):
        for name, (_, envs) in VARIANTS.items():
            lib = os.path.join(EXP, "libvbmc_hip_%s.so" % name)
            if not os.path.exists(lib):
                continue
            for e in envs:
                tag = name + "".join(" %s=%s" % kv for kv in e.items())
                dump = os.path.join(tmp, "out_%s_%d.npz" % (name, rep))
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "one", str(R), str(Ns), dump],
                                     env=dict(os.environ, VBMC_HIP_LIB=lib, **e), capture_output=True, text=True)
                line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
                if tag not in tags:
                    tags.append(tag)
                if not line:
                    notes[tag] = "FAILED " + out.stderr[-400:]
                    continue
                runs.setdefault(tag, []).append(json.loads(line[-1]))
                z = np.load(dump)
                if ref is None:
                    ref = z

                def rel(a, b):
                    return float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b))))

                notes[tag] = "dH %.1e dHgrad %.1e" % (rel(z["H"], ref["H"]), rel(z["dH"], ref["dH"]))
    base = basemin = None
    for tag in tags:
        if tag not in runs:
            print("%-28s %s" % (tag, notes.get(tag)))
            continue
        ms = float(np.median([r_["ms"] for r_ in runs[tag]]))
        mn = float(np.min([r_["min"] for r_ in runs[tag]]))
        if base is None:
            base = ms
        extra = ""
        if "eps_ms" in runs[tag][0]:
            extra = "   eps-from-memory %.3f ms" % float(np.median([r_["eps_ms"] for r_ in runs[tag]]))
        if basemin is None:
            basemin = mn
        print("%-22s median %.3f ms %+5.1f %%   min %.3f ms %+5.1f %%   (runs %s)   %s%s" % (tag, ms, 100 * (ms - base) / base, mn, 100 * (mn - basemin) / basemin, " ".join("%.3f" % r_["ms"] for r_ in runs[tag]), notes[tag], extra))


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "build"
    if cmd == "build":
        build(int(os.environ.get("ENT_AB_QS", "3")))       # ENT_AB_QS=n: the translation unit of QS = n (D = 4 n - 5 .. 4 n - 2)
    elif cmd == "one":
        one(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else None)
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 64, int(sys.argv[3]) if len(sys.argv) > 3 else 10000)
