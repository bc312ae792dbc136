"""CPU: libvbmc_hip.so loads and exports every symbol include/vbmc_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if not fn.endswith(".h"):
            continue
        txt = open(os.path.join(ROOT, "include", fn)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        for m in re.finditer(r"\b(vbmc_[a-z0-9_]+)\s*\(", txt):
            names.add(m.group(1))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g

    g.build()
    from vbmc_amd import _lib

    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 10
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    hdr = open(os.path.join(ROOT, "include", "vbmc_hip.h")).read()
    ver = int(re.search(r"#define\s+VBMC_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert lib.vbmc_abi_version() == ver == _lib.ABI_VERSION


def test_no_cpu_fallback_without_gpu():
    """Without a gfx950 device the product path must fail loudly, never compute on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import vbmc_amd

    with pytest.raises(vbmc_amd.VbmcHipError):
        vbmc_amd.Context(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "vbmc_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in txt.replace("no CPU fallback", ""), os.path.join(dirpath, fn)


def test_entropy_plan_reporting_hook():
    """vbmc_entropy_plan is a pure host function (no device needed): the instantiation the library's policy picks.  Locks the
    choices the profiles were taken with: the headline shape runs three k-tiles + a two-component tail on one wave, configs[4]
    three k-tiles + tail on two waves, K = 64 four full k-tiles, D > 34 the VALU kernel."""
    import bench

    assert bench.entropy_kernel_label(10, 50) == "k_entropy_mfma<QS=3,KT=3+tail2,grad>"
    assert bench.entropy_kernel_label(20, 100) == "k_entropy_mfma<QS=6,KT=3+tail2,grad,HV=2>"
    assert bench.entropy_kernel_label(10, 64) == "k_entropy_mfma<QS=3,KT=4,grad>"
    assert bench.entropy_kernel_label(10, 56) == "k_entropy_mfma<QS=3,KT=3+tail8,grad>"
    assert bench.entropy_kernel_label(20, 56) == "k_entropy_mfma<QS=6,KT=4,grad>"          # two values per lane would spill there
    assert bench.entropy_kernel_label(6, 10) == "k_entropy_lane<DT=6,KP=10,grad> (+ log-joint role)"      # the small class (round 6)
    assert bench.entropy_kernel_label(13, 10) == "k_entropy_mfma<QS=4,KT=1,grad>"
    assert bench.entropy_kernel_label(6, 17) == "k_entropy_mfma<QS=2,KT=1+tail1,grad>"
    assert bench.entropy_kernel_label(10, 200) == "k_entropy_mfma<QS=3,KT=3+tail2,grad,HV=4>"
    assert bench.entropy_kernel_label(40, 10).startswith("k_entropy<D=40")


def test_limits_are_the_librarys_and_the_shims_ask_for_them():
    """vbmc_get_limits (a host function) reports what the validation enforces -- the numbers tests/test_gpu_limits.py exercises on the
    device (K = 400 of 512, N = 4500 of 9696) -- and the MATLAB side takes them from there: no shim restates a limit."""
    import ctypes as C
    import glob
    import re

    from vbmc_amd import _lib

    class Lim(C.Structure):
        _fields_ = [("struct_size", C.c_uint32)] + [(n, C.c_int32) for n in ("max_D", "max_K", "max_N", "max_Na", "max_T_vargrad", "delta_ok", "meanfun_mask")]

    lib = _lib.load()
    lib.vbmc_get_limits.argtypes = [C.POINTER(Lim)]
    lib.vbmc_get_limits.restype = C.c_int
    lim = Lim()
    lim.struct_size = C.sizeof(Lim)
    assert lib.vbmc_get_limits(C.byref(lim)) == 0
    assert (lim.max_D, lim.max_K, lim.max_N, lim.max_Na, lim.delta_ok, lim.meanfun_mask) == (32, 512, 9696, 256, 1, 0b10011)
    assert 3800 <= lim.max_T_vargrad <= 4100          # ("about 4000": five T-vectors beside the reduction scratch in 160 KB)
    bad = Lim()
    bad.struct_size = 4
    assert lib.vbmc_get_limits(C.byref(bad)) != 0
    # the device tests reach into the upper half of each range
    txt = open(os.path.join(ROOT, "tests", "test_gpu_limits.py")).read()
    ks = [int(x) for x in re.findall(r"\(\d+, \d+, (\d+), \d, \d+\)", txt)]
    assert max(ks) > lim.max_K // 2 and max(ks) <= lim.max_K
    assert "4500" in txt and 4500 <= lim.max_N
    # no .m file carries the numbers (they drifted once: K <= 256 / N <= 3872 in round 5's shims against 512 / 9696 in the library)
    for f in glob.glob(os.path.join(ROOT, "matlab", "*.m")):
        code = "\n".join(ln.split("%")[0] for ln in open(f).read().split("\n"))
        assert not re.search(r"\bK\s*(<=|>)\s*\d{3}|\bD\s*(<=|>)\s*32\b|size\(gp\.X,1\)\s*<=\s*\d+", code), f
    sup = open(os.path.join(ROOT, "matlab", "vbmc_hip_supported.m")).read()
    assert "vbmc_hip_mex('limits')" in sup and "lim.max_K" in sup and "lim.max_N" in sup and "lim.delta_ok" in sup
