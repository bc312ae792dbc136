"""Where does the time of k_entropy_mfma go?  Builds variants of libvbmc_hip.so whose entropy kernel has one phase switched
off (-DVBMC_EXP_NO...: results meaningless, durations comparable) and times the headline launch with each.

    python tools/ent_experiments.py build            (here: cross-compiles vbmc_amd/lib/exp/libvbmc_hip_<name>.so)
    python tools/ent_experiments.py run  [R] [Ns]    (GPU box: HIP-event duration of the entropy kernel per variant)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "vbmc_amd", "lib", "exp")
OBJ = os.path.join(ROOT, "vbmc_amd", "lib", "obj")
VARIANTS = {"base": [], "norng": ["-DVBMC_EXP_NORNG"], "noexp": ["-DVBMC_EXP_NOEXP"], "nopv": ["-DVBMC_EXP_NOPV"],
            "noepi": ["-DVBMC_EXP_NOEPI"], "now": ["-DVBMC_EXP_NOW"], "nos": ["-DVBMC_EXP_NOS"],
            "nomfma": ["-DVBMC_EXP_NOS", "-DVBMC_EXP_NOPV"],
            "novalu": ["-DVBMC_EXP_NORNG", "-DVBMC_EXP_NOEXP", "-DVBMC_EXP_NOEPI", "-DVBMC_EXP_NOW"],
            "stag": ["-DVBMC_STAG"], "now_stag": ["-DVBMC_EXP_NOW", "-DVBMC_STAG"],   # the staggered two-sign schedule at four k-tiles too
            # two waves per SIMD (256 VGPRs, no spills) for every two-k-tile kernel with a component tail
            "x_w2": ["-DVBMC_ENT_WAVES(KT_,QS_,TL_,HV_)=((((KT_)<=2&&(QS_)<=4)&&!((KT_)==2&&(TL_)))?3:2)"],
            "bare": ["-DVBMC_EXP_NORNG", "-DVBMC_EXP_NOEXP", "-DVBMC_EXP_NOEPI", "-DVBMC_EXP_NOW", "-DVBMC_EXP_NOS", "-DVBMC_EXP_NOPV"]}


def build(qs=3):
    only = os.environ.get("VBMC_EXP_ONLY")          # comma-separated subset of the variants
    if only:
        for k in list(VARIANTS):
            if k not in only.split(","):
                del VARIANTS[k]
    os.makedirs(EXP, exist_ok=True)
    others = [os.path.join(OBJ, "vbmc_hip.o")] + [os.path.join(OBJ, "ent_mfma_qs%d.o" % q) for q in range(1, 10) if q != qs]
    procs = []
    for name, flags in VARIANTS.items():
        o = os.path.join(EXP, "ent_%s.o" % name)
        procs.append((name, o, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
                                                 "-Wno-pass-failed", "-DQS_VALUE=%d" % qs] + flags +
                                                ["-c", os.path.join(ROOT, "vbmc_amd", "csrc", "ent_mfma_inst.hip"), "-o", o])))
    for name, o, p in procs:
        assert p.wait() == 0, name
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", o] + others +
                              ["-o", os.path.join(EXP, "libvbmc_hip_%s.so" % name)])
        os.remove(o)
    print("built", sorted(VARIANTS))


def one(R, Ns):
    sys.path.insert(0, ROOT)
    import numpy as np

    import vbmc_amd
    from bench import synth_inputs

    D, N, K, S = 10, 400, int(os.environ.get('EXP_K', '50')), 20
    inp = synth_inputs(0, D, N, K, S)
    eng = vbmc_amd.Engine(0)
    gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng)
    vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
    vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
    theta0 = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
    th = np.asfortranarray(theta0[:, None] + 0.05 * np.random.default_rng(100).standard_normal((theta0.size, R)))
    for i in range(3):
        vbmc_amd.negelcbo_batch(th, 0, vp, gp, Ns, True, 0, seed=i, engine=eng, outputs=("F",))
    eng.ctx.set_profiling(True)
    ms = []
    for i in range(8):
        vbmc_amd.negelcbo_batch(th, 0, vp, gp, Ns, True, 0, seed=10 + i, engine=eng, outputs=("F",))
        ms.append(eng.ctx.last_kernel_ms()[0])
    print(json.dumps({"ms": float(np.median(ms))}))


def run(R, Ns):
    res = {}
    for name in VARIANTS:
        lib = os.path.join(EXP, "libvbmc_hip_%s.so" % name)
        if not os.path.exists(lib):
            continue
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "one", str(R), str(Ns)], env=dict(os.environ, VBMC_HIP_LIB=lib),
                             capture_output=True, text=True)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        res[name] = json.loads(line[-1])["ms"] if line else out.stderr[-300:]
    base = res.get("base")
    for k, v in res.items():
        print("%-8s %s" % (k, ("%.3f ms  (%+.1f %%)" % (v, 100 * (v - base) / base)) if isinstance(v, float) and isinstance(base, float) else v))


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "build"
    if cmd == "build":
        build()
    else:
        R = int(sys.argv[2]) if len(sys.argv) > 2 else 64
        Ns = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
        (one if cmd == "one" else run)(R, Ns)
