"""Harness that EXECUTES the MEX gateway (matlab/vbmc_hip_mex.cpp) without MATLAB: the gateway is compiled against the
functional mock of the mx* / mex* API under tests/mock_mex/ and linked with libvbmc_hip.so into one shared object;
``Mex.call(nlhs, cmd, *args)`` builds the prhs array from Python values the way MATLAB would hand them over, calls
mexFunction and converts plhs back.  Test infrastructure only.

Python -> MATLAB value mapping: float / int -> 1x1 double; numpy float64 array -> double array of the same shape
(column-major; 1-D arrays become columns); numpy uint64 scalar -> uint64 scalar (device handles); bool -> logical scalar;
str -> char row; None -> 0x0 double; dict -> 1x1 struct; list of dicts -> 1xn struct array."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "mock_mex")
OUT = os.path.join(MOCK, "_build", "libvbmc_hip_mex_mock.so")
SRCS = [os.path.join(ROOT, "matlab", "vbmc_hip_mex.cpp"), os.path.join(MOCK, "mock_mx.cpp")]
DEPS = SRCS + [os.path.join(MOCK, "mex.h"), os.path.join(ROOT, "include", "vbmc_hip.h")]
LIBDIR = os.path.join(ROOT, "vbmc_amd", "lib")

mxDOUBLE, mxLOGICAL, mxCHAR, mxSTRUCT, mxUINT8, mxINT32, mxUINT64 = 6, 3, 4, 2, 9, 12, 15
_NP = {mxDOUBLE: np.float64, mxUINT8: np.uint8, mxINT32: np.int32, mxUINT64: np.uint64, mxLOGICAL: np.bool_}


class MexError(RuntimeError):
    """What MATLAB would see as an MException: .identifier and .message."""

    def __init__(self, identifier, message):
        super().__init__("%s: %s" % (identifier, message))
        self.identifier = identifier
        self.message = message


def build(force=False):
    """g++ the gateway + mock into tests/mock_mex/_build/ (in-tree: it travels to the GPU box with the snapshot)."""
    lib = os.path.join(LIBDIR, "libvbmc_hip.so")
    if not os.path.exists(lib):
        raise ImportError("build libvbmc_hip.so first (__graft_entry__.build())")
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in DEPS + [lib]):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-fPIC", "-shared"] + SRCS + [
        "-I", MOCK, "-I", os.path.join(ROOT, "include"), "-L", LIBDIR, "-lvbmc_hip", "-Wl,-rpath," + LIBDIR, "-o", OUT]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("mock MEX build failed:\n" + r.stderr)
    return OUT


class Mex:
    def __init__(self):
        self.lib = C.CDLL(build())
        L = self.lib
        vp, sz = C.c_void_p, C.c_size_t
        L.mxCreateDoubleMatrix.restype = vp; L.mxCreateDoubleMatrix.argtypes = [sz, sz, C.c_int]
        L.mxCreateNumericArray.restype = vp; L.mxCreateNumericArray.argtypes = [sz, C.POINTER(sz), C.c_int, C.c_int]
        L.mxCreateLogicalScalar.restype = vp; L.mxCreateLogicalScalar.argtypes = [C.c_bool]
        L.mxCreateString.restype = vp; L.mxCreateString.argtypes = [C.c_char_p]
        L.mxCreateStructMatrix.restype = vp; L.mxCreateStructMatrix.argtypes = [sz, sz, C.c_int, C.POINTER(C.c_char_p)]
        L.mxSetField.restype = None; L.mxSetField.argtypes = [vp, sz, C.c_char_p, vp]
        L.mxDestroyArray.restype = None; L.mxDestroyArray.argtypes = [vp]
        L.mxGetData.restype = vp; L.mxGetData.argtypes = [vp]
        L.mxGetClassID.restype = C.c_int; L.mxGetClassID.argtypes = [vp]
        L.mxGetNumberOfDimensions.restype = sz; L.mxGetNumberOfDimensions.argtypes = [vp]
        L.mxGetDimensions.restype = C.POINTER(sz); L.mxGetDimensions.argtypes = [vp]
        L.mxGetNumberOfElements.restype = sz; L.mxGetNumberOfElements.argtypes = [vp]
        L.mxGetField.restype = vp; L.mxGetField.argtypes = [vp, sz, C.c_char_p]
        L.mxGetNumberOfFields.restype = C.c_int; L.mxGetNumberOfFields.argtypes = [vp]
        L.mxGetFieldNameByNumber.restype = C.c_char_p; L.mxGetFieldNameByNumber.argtypes = [vp, C.c_int]
        L.mock_mex_call.restype = C.c_int
        L.mock_mex_call.argtypes = [C.c_int, C.POINTER(vp), C.c_int, C.POINTER(vp), C.c_char_p, sz, C.c_char_p, sz]
        L.mock_mex_live_arrays.restype = C.c_long
        L.mock_mex_lock_count.restype = C.c_int
        L.mock_mex_run_at_exit.restype = None

    # ---- Python -> mxArray
    def to_mx(self, v):
        L = self.lib
        if v is None:
            return L.mxCreateDoubleMatrix(0, 0, 0)
        if isinstance(v, (bool, np.bool_)):
            return L.mxCreateLogicalScalar(bool(v))
        if isinstance(v, str):
            return L.mxCreateString(v.encode())
        if isinstance(v, dict):
            return self._struct([v])
        if isinstance(v, (list, tuple)) and v and isinstance(v[0], dict):
            return self._struct(list(v))
        if isinstance(v, np.uint64):
            return self._numeric(np.array([[v]], dtype=np.uint64), mxUINT64)
        a = np.asarray(v)
        if a.dtype == np.uint64:                      # handle vectors keep their class, as in MATLAB
            return self._numeric(a.reshape(1, -1) if a.ndim < 2 else a, mxUINT64)
        if a.dtype == np.bool_ and a.size == 1:
            return L.mxCreateLogicalScalar(bool(a.reshape(-1)[0]))
        a = np.asarray(a, dtype=np.float64)
        if a.ndim == 0:
            a = a.reshape(1, 1)
        elif a.ndim == 1:
            a = a.reshape(-1, 1)
        return self._numeric(a, mxDOUBLE)

    def _numeric(self, a, cls):
        dims = (C.c_size_t * a.ndim)(*a.shape)
        h = self.lib.mxCreateNumericArray(a.ndim, dims, cls, 0)
        if a.size:
            buf = np.asfortranarray(a)
            C.memmove(self.lib.mxGetData(h), buf.ctypes.data, buf.nbytes)
        return h

    def _struct(self, elems):
        names = list(elems[0].keys())
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        h = self.lib.mxCreateStructMatrix(1, len(elems), len(names), arr)
        for i, e in enumerate(elems):
            assert list(e.keys()) == names, "struct array elements must share their fields"
            for n in names:
                self.lib.mxSetField(h, i, n.encode(), self.to_mx(e[n]))
        return h

    # ---- mxArray -> Python
    def from_mx(self, h):
        if not h:
            return None
        L = self.lib
        cls = L.mxGetClassID(h)
        nd = L.mxGetNumberOfDimensions(h)
        dims = tuple(L.mxGetDimensions(h)[i] for i in range(nd))
        n = L.mxGetNumberOfElements(h)
        if cls == mxSTRUCT:        # 1 x 1 struct -> dict (what 'limits' returns)
            assert n == 1
            out = {}
            for i in range(L.mxGetNumberOfFields(h)):
                name = L.mxGetFieldNameByNumber(h, i).decode()
                out[name] = self.from_mx(L.mxGetField(h, 0, name.encode()))
            return out
        dt = _NP[cls]
        out = np.empty(dims, dtype=dt, order="F")
        if n:
            C.memmove(out.ctypes.data, L.mxGetData(h), out.nbytes)
        return out

    # ---- one invocation, MATLAB style
    def call(self, nlhs, cmd, *args):
        """[out1, ..., out_nlhs] = vbmc_hip_mex(cmd, args...).  Returns a list of nlhs numpy arrays (at least one slot, like
        MATLAB's ``ans``; None where the gateway left an output unassigned -- MATLAB would raise for those)."""
        prhs_v = [self.to_mx(cmd)] + [self.to_mx(a) for a in args]
        nrhs = len(prhs_v)
        prhs = (C.c_void_p * nrhs)(*prhs_v)
        nout = max(nlhs, 1)
        plhs = (C.c_void_p * nout)()
        eid, emsg = C.create_string_buffer(128), C.create_string_buffer(1024)
        rc = self.lib.mock_mex_call(nlhs, plhs, nrhs, prhs, eid, 128, emsg, 1024)
        try:
            if rc:
                raise MexError(eid.value.decode(), emsg.value.decode())
            return [self.from_mx(plhs[i]) for i in range(nout)][: max(nlhs, 1)]
        finally:
            for i in range(nout):
                if plhs[i]:
                    self.lib.mxDestroyArray(plhs[i])
            for h in prhs_v:
                self.lib.mxDestroyArray(h)

    def live_arrays(self):
        return self.lib.mock_mex_live_arrays()

    def close(self):
        """clear mex: the registered mexAtExit handler (destroys the gateway's device context)."""
        self.lib.mock_mex_run_at_exit()


_mex = None


def mex():
    global _mex
    if _mex is None:
        _mex = Mex()
    return _mex
