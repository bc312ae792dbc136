"""Builds variants of the library with the entropy kernel's tuning knobs set (-DVBMC_TUNE_PV2=0/1 -DVBMC_TUNE_EVREG=0/1) into
vbmc_amd/lib/tune/lib_<pv2><evreg>.so (all nine QS translation units each).  For tools/tune_sweep.sh."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "vbmc_amd", "lib", "tune")
OBJ = os.path.join(ROOT, "vbmc_amd", "lib", "obj")
os.makedirs(OUT, exist_ok=True)
variants = sys.argv[1:] or ["00", "01", "10", "11"]
procs = []
for v in variants:
    for q in range(1, 10):
        o = os.path.join(OUT, "ent_%s_qs%d.o" % (v, q))
        procs.append((v, o, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
                                              "-Wno-pass-failed", "-DQS_VALUE=%d" % q, "-DVBMC_TUNE_PV2=%s" % v[0], "-DVBMC_TUNE_EVREG=%s" % v[1],
                                              "-c", os.path.join(ROOT, "vbmc_amd", "csrc", "ent_mfma_inst.hip"), "-o", o])))
    for _, o, p in [x for x in procs if x[0] == v]:
        assert p.wait() == 0, o
    objs = [o for vv, o, _ in procs if vv == v]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(OBJ, "vbmc_hip.o")] + objs +
                          ["-o", os.path.join(OUT, "lib_%s.so" % v)])
    for o in objs:
        os.remove(o)
    print("built", v, flush=True)
