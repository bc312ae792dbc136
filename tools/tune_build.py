"""Builds variants of the library with extra -D flags on the entropy-kernel translation units (all nine QS) into
vbmc_amd/lib/tune/lib_<name>.so.   usage: python tools/tune_build.py base: stag:-DVBMC_STAG pv2:-DVBMC_TUNE_PV2=1,-DVBMC_TUNE_EVREG=0
For tools/tune_sweep.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "vbmc_amd", "lib", "tune")
OBJ = os.path.join(ROOT, "vbmc_amd", "lib", "obj")
os.makedirs(OUT, exist_ok=True)
variants = [(v.split(":", 1)[0], [f for f in v.split(":", 1)[1].split(",") if f]) for v in (sys.argv[1:] or ["base:"])]
for name, flags in variants:
    # "@DIR" among the flags: the entropy translation unit of another source tree (e.g. `git archive HEAD vbmc_amd/csrc include | tar -x -C DIR`)
    src_root = ([f[1:] for f in flags if f.startswith("@")] or [ROOT])[0]
    flags = [f for f in flags if not f.startswith("@")]
    procs = []
    for q in range(1, 10):
        o = os.path.join(OUT, "ent_%s_qs%d.o" % (name, q))
        procs.append((o, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
                                           "-Wno-pass-failed", "-DQS_VALUE=%d" % q] + flags +
                                          ["-c", os.path.join(src_root, "vbmc_amd", "csrc", "ent_mfma_inst.hip"), "-o", o])))
    for o, p in procs:
        assert p.wait() == 0, o
    objs = [o for o, _ in procs]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(OBJ, "vbmc_hip.o")] + objs +
                          ["-ldl", "-o", os.path.join(OUT, "lib_%s.so" % name)])
    for o in objs:
        os.remove(o)
    print("built", name, flags, flush=True)
