"""Phase timeline of k_finalize_ws inside the single-chain Adam loop: a build of the library with -DVBMC_INSTRUMENT
stamps the 100 MHz counter at the kernel's phases (restart 0's workgroup; the nine single-wave tasks each stamp their own start / end).

    python tools/fin_timeline.py build     (here: vbmc_amd/lib/exp/libvbmc_hip_finclk.so)
    python tools/fin_timeline.py run       (GPU box: the last iteration's stamps of a 60-iteration chain at D = 10, N = 400, K = 50, Ns = 28)
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "vbmc_amd", "lib", "exp")
OBJ = os.path.join(ROOT, "vbmc_amd", "lib", "obj")
LIB = os.path.join(EXP, "libvbmc_hip_finclk.so")


def build():
    os.makedirs(EXP, exist_ok=True)
    o = os.path.join(EXP, "vbmc_hip_finclk.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-DVBMC_INSTRUMENT",
                           "-c", os.path.join(ROOT, "vbmc_amd", "csrc", "vbmc_hip.hip"), "-o", o])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", o] +
                          [os.path.join(OBJ, "ent_mfma_qs%d.o" % q) for q in range(1, 10)] + ["-ldl", "-o", LIB])
    os.remove(o)
    print("built", LIB)


def run():
    os.environ["VBMC_HIP_LIB"] = LIB
    os.environ.setdefault("PROF_NS", "28")
    sys.path.insert(0, ROOT)
    sys.argv = [sys.argv[0]]
    import runpy

    runpy.run_path(os.path.join(ROOT, "tools", "prof_adam.py"), run_name="__main__")
    from vbmc_amd import _lib

    lib = _lib.load()
    buf = (C.c_ulonglong * 64)()
    assert lib.vbmc_dbg_fin_read(buf) == 0
    t = [int(x) for x in buf]
    us = lambda a, b: (t[b] - t[a]) / 100.0  # noqa: E731
    print("k_finalize_ws phases, last iteration [us] (100 MHz counter):")
    print("  staging + zeroing + barrier        %6.2f" % us(0, 1))
    print("  single-wave tasks + barrier        %6.2f" % us(1, 2))
    names = ["G + small blocks", "dG mu", "H + dH sigma", "dH mu", "dH lambda", "dH eta (K x K)", "bounds mu", "bounds lnscale", "bounds eta + weights",
             "dG lambda", "bounds lnscale: lambda"]
    for k in range(11):
        print("    task %d %-22s start +%5.2f  duration %5.2f" % (k, names[k], (t[16 + 2 * k] - t[1]) / 100.0, (t[17 + 2 * k] - t[16 + 2 * k]) / 100.0))
    print("  assembly + Adam update             %6.2f" % us(2, 3))
    print("  barrier, theta through LDS, barrier%6.2f" % us(3, 4))
    print("  unpacking the next theta (prep)    %6.2f" % us(4, 5))
    print("  total                              %6.2f" % us(0, 5))


if __name__ == "__main__":
    (build if (len(sys.argv) < 2 or sys.argv[1] == "build") else run)()
