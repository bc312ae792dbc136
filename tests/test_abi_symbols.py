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
