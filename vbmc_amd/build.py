"""Build libvbmc_hip.so (and the microbenchmark) for gfx950 with hipcc, in-tree.

Translation units are compiled in parallel: vbmc_hip.hip (ABI + all kernels but the MFMA entropy
family), ent_mfma_inst.hip once per QS = 1..9 (k_entropy_mfma<QS, KT, grad> for every KT) and
ent_lane_inst.hip once per padded dimension DT = 2, 4, .., 12 (k_entropy_lane<DT, KP, grad>).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "lib", "obj")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
MAIN_DEPS = ["vbmc_hip.hip", "abi_elbo.hip", "abi_gp.hip", "abi_comm.hip", "common.h", "device_math.h", "exp2_tab1k.h", "elbo_types.h", "elbo_kernels.h", "logjoint_body.h", "trsm_mfma.h",
             "var_kernels.h", "gp_kernels.h", "chol_mfma.h", os.path.join("..", "..", "include", "vbmc_hip.h")]
MFMA_DEPS = ["ent_mfma_inst.hip", "entropy_mfma.h", "device_math.h", "exp2_tab1k.h", "elbo_types.h", "logjoint_body.h"]
LANE_DEPS = ["ent_lane_inst.hip", "entropy_lane.h", "device_math.h", "exp2_tab1k.h", "elbo_types.h", "logjoint_body.h"]
QS_RANGE = range(1, 10)
LANE_DT = (2, 4, 6, 8, 10, 12)


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    jobs = []
    objs = []
    main_o = os.path.join(OBJDIR, "vbmc_hip.o")
    objs.append(main_o)
    if force or _newer(main_o, [os.path.join(CSRC, d) for d in MAIN_DEPS]):
        jobs.append([hipcc] + FLAGS + ["-c", os.path.join(CSRC, "vbmc_hip.hip"), "-o", main_o])
    for qs in QS_RANGE:
        o = os.path.join(OBJDIR, "ent_mfma_qs%d.o" % qs)
        objs.append(o)
        if force or _newer(o, [os.path.join(CSRC, d) for d in MFMA_DEPS]):
            jobs.append([hipcc] + FLAGS + ["-DQS_VALUE=%d" % qs, "-c", os.path.join(CSRC, "ent_mfma_inst.hip"), "-o", o])
    for dt in LANE_DT:
        o = os.path.join(OBJDIR, "ent_lane_dt%d.o" % dt)
        objs.append(o)
        if force or _newer(o, [os.path.join(CSRC, d) for d in LANE_DEPS]):
            jobs.append([hipcc] + FLAGS + ["-DDT_VALUE=%d" % dt, "-c", os.path.join(CSRC, "ent_lane_inst.hip"), "-o", o])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(lambda c: _run(c, verbose), jobs))
    lib = os.path.join(LIBDIR, "libvbmc_hip.so")
    if force or jobs or _newer(lib, objs):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", lib], verbose)
    mb_src = os.path.join(ROOT, "tools", "microbench.hip")
    mb = os.path.join(LIBDIR, "microbench")
    if os.path.exists(mb_src) and (force or _newer(mb, [mb_src, os.path.join(CSRC, "device_math.h")])):
        _run([hipcc] + FLAGS[:3] + ["-Wno-unused-value", mb_src, "-o", mb], verbose)
    # stand-alone harness of the Cholesky kernel (profiles/r02_chol.md): built with the phase stamps compiled in
    cb_src = os.path.join(ROOT, "tools", "chol_bench.hip")
    cb = os.path.join(LIBDIR, "chol_bench")
    cb_deps = [cb_src] + [os.path.join(CSRC, d) for d in ("chol_mfma.h", "gp_kernels.h", "var_kernels.h", "trsm_mfma.h", "common.h", "device_math.h")]
    if os.path.exists(cb_src) and (force or _newer(cb, cb_deps)):
        _run([hipcc] + FLAGS[:3] + ["-Wno-unused-value", "-DCHOL_TS", "-I" + os.path.join(ROOT, "include"), cb_src, "-o", cb], verbose)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv)
