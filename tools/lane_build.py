"""Builds variants of the library with extra -D flags on the lane-entropy translation units into vbmc_amd/lib/tune/lib_<name>.so
(VBMC_HIP_LIB selects one).   usage: python tools/lane_build.py inst:-DVBMC_INSTRUMENT occ3:-DENT_LANE_OCC\\(a,b\\)=3"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "vbmc_amd", "lib", "tune")
OBJ = os.path.join(ROOT, "vbmc_amd", "lib", "obj")
os.makedirs(OUT, exist_ok=True)
variants = [(v.split(":", 1)[0], [f for f in v.split(":", 1)[1].split(",") if f]) for v in (sys.argv[1:] or ["base:"])]
DTS = (2, 4, 6, 8, 10, 12)
for name, flags in variants:
    procs = []
    for dt in DTS:
        o = os.path.join(OUT, "lane_%s_dt%d.o" % (name, dt))
        procs.append((o, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
                                           "-Wno-pass-failed", "-DDT_VALUE=%d" % dt] + flags +
                                          ["-c", os.path.join(ROOT, "vbmc_amd", "csrc", "ent_lane_inst.hip"), "-o", o])))
    for o, p in procs:
        assert p.wait() == 0, o
    objs = [o for o, _ in procs]
    base = [os.path.join(OBJ, "vbmc_hip.o")] + [os.path.join(OBJ, "ent_mfma_qs%d.o" % q) for q in range(1, 10)]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + base + objs +
                          ["-ldl", "-o", os.path.join(OUT, "lib_%s.so" % name)])
    for o in objs:
        os.remove(o)
    print("built", name, flags, flush=True)
