"""CPU: matlab/vbmc_hip_mex.cpp type-checks against include/vbmc_hip.h (with a mock mex.h: MATLAB is absent here) and every
vbmc_* symbol it references is exported by libvbmc_hip.so -- catches drift between the C ABI and the MEX gateway.  Also a
static check that the .m shims only use MEX commands the gateway implements, with argument counts it accepts."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEX = os.path.join(ROOT, "matlab", "vbmc_hip_mex.cpp")


def test_gateway_compiles_against_the_abi(tmp_path):
    import __graft_entry__ as g

    g.build()
    obj = str(tmp_path / "vbmc_hip_mex.o")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror=return-type", "-fPIC", "-c", MEX,
                        "-I", os.path.join(ROOT, "tests", "mock_mex"), "-I", os.path.join(ROOT, "include"), "-o", obj],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    und = subprocess.run(["nm", "-u", obj], capture_output=True, text=True).stdout
    wanted = sorted(set(re.findall(r"\b(vbmc_[a-z0-9_]+)\b", und)))
    assert len(wanted) >= 15, wanted
    from vbmc_amd import _lib

    exp = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\b(vbmc_[a-z0-9_]+)\b", exp))
    assert not [w for w in wanted if w not in exported]


def test_no_long_jump_with_live_cpp_objects():
    """mexErrMsgIdAndTxt long-jumps without running destructors: it may only be called from mexFunction itself."""
    src = open(MEX).read()
    body = src[src.index("static int dispatch("):src.index("void mexFunction(")]
    assert "mexErrMsg" not in body
    assert src.count("mexErrMsgIdAndTxt(") == 1 + src[: src.index("#include")].count("mexErrMsgIdAndTxt(")


def test_shims_use_only_commands_the_gateway_implements():
    src = open(MEX).read()
    cmds = set(re.findall(r'!strcmp\(cmd, "([a-z_0-9]+)"\)', src))
    used = set()
    for fn in os.listdir(os.path.join(ROOT, "matlab")):
        if fn.endswith(".m"):
            used |= set(re.findall(r"vbmc_hip_mex\('([a-z_0-9]+)'", open(os.path.join(ROOT, "matlab", fn)).read()))
    assert used and used <= cmds, used - cmds


def test_gateway_links_and_runs_against_the_functional_mock():
    """The gateway + tests/mock_mex/mock_mx.cpp + libvbmc_hip.so link into one object and mexFunction EXECUTES (tests/_mex.py):
    without a GPU every command stops at the context ('vbmc_hip:nodevice', raised through the mock's mexErrMsgIdAndTxt), usage
    errors are reported before that, and nothing a failed call created stays alive.  The command-by-command comparison with
    the ctypes path is tests/test_gpu_mex.py."""
    import numpy as np
    import pytest

    from tests import _mex

    m = _mex.mex()
    with pytest.raises(_mex.MexError) as e:
        m.call(0, 5.0)
    assert e.value.identifier == "vbmc_hip:usage"
    import torch

    if not torch.cuda.is_available():
        with pytest.raises(_mex.MexError) as e:
            m.call(1, "sq_dist", np.zeros((2, 3)))
        assert e.value.identifier == "vbmc_hip:nodevice"
    assert m.live_arrays() == 0
    # the mock's own value mapping: what MATLAB would hand the gateway, read back unchanged
    a = np.arange(24, dtype=np.float64).reshape(2, 3, 4)
    h = m.to_mx(a)
    assert np.array_equal(m.from_mx(h), a)
    m.lib.mxDestroyArray(h)
    h = m.to_mx([{"x": 1.0, "s": "ab"}, {"x": np.ones((2, 2)), "s": None}])
    assert m.live_arrays() == 5
    m.lib.mxDestroyArray(h)
    assert m.live_arrays() == 0
