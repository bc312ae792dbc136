"""Build libvbmc_hip.so (and the microbenchmark) for gfx950 with hipcc, in-tree."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
SOURCES = ["vbmc_hip.hip"]
HEADERS = ["abi_elbo.hip", "abi_gp.hip", "common.h", "device_math.h", "elbo_kernels.h", "var_kernels.h", "gp_kernels.h", "entropy_mfma.h", os.path.join("..", "..", "include", "vbmc_hip.h")]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS]
    lib = os.path.join(LIBDIR, "libvbmc_hip.so")
    if force or _newer(lib, deps):
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value"] + srcs + ["-o", lib]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    mb_src = os.path.join(ROOT, "tools", "microbench.hip")
    mb = os.path.join(LIBDIR, "microbench")
    if os.path.exists(mb_src) and (force or _newer(mb, [mb_src, os.path.join(CSRC, "device_math.h")])):
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", mb_src, "-o", mb]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv)
