"""ctypes binding of libvbmc_hip.so (include/vbmc_hip.h).

The product path has NO CPU fallback: importing this module without the built
library, or creating a Context without a gfx950 device, raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

# (GPU_MAX_HW_QUEUES: raised by the library itself, inside vbmc_ctx_create ahead of its first HIP call -- round 5; importing this module no
# longer touches the process environment.  An application that initialises another HIP user first -- bench.py imports torch -- sets it
# itself.)

_HERE = os.path.dirname(os.path.abspath(__file__))
# VBMC_HIP_LIB: an alternative build of the SAME library (kernel A/B experiments, tools/ent_experiments.py)
LIB_PATH = os.environ.get("VBMC_HIP_LIB") or os.path.join(_HERE, "lib", "libvbmc_hip.so")

ABI_VERSION = 5   # include/vbmc_hip.h: VBMC_ABI_VERSION
VBMC_OK, VBMC_ERR_INVALID, VBMC_ERR_NO_DEVICE, VBMC_ERR_HIP, VBMC_ERR_UNSUPPORTED, VBMC_ERR_NOT_POSDEF = range(6)
_STATUS_NAMES = {0: "OK", 1: "INVALID", 2: "NO_DEVICE", 3: "HIP", 4: "UNSUPPORTED", 5: "NOT_POSDEF"}


class VbmcHipError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("libvbmc_hip: %s: %s" % (_STATUS_NAMES.get(status, status), msg))
        self.status = status
        self.message = msg


class VbmcUnsupported(VbmcHipError):
    """Option outside the accelerated path; the caller should fall through to the reference .m."""


_dp = C.POINTER(C.c_double)


class ElboArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("D", C.c_int32), ("K", C.c_int32), ("R", C.c_int32),
        ("optimize", C.c_int32 * 4),
        ("theta", _dp),
        ("vp_mu", _dp), ("vp_sigma", _dp), ("vp_lambda", _dp), ("vp_w", _dp), ("vp_delta", _dp),
        ("Ns", C.c_int32),
        ("eps_mode", C.c_int32),
        ("eps", C.c_void_p),
        ("eps_shared", C.c_int32),
        ("seed", C.c_uint64),
        ("compute_grad", C.c_int32), ("compute_var", C.c_int32), ("separate_K", C.c_int32),
        ("beta", C.c_double),
        ("bnd_lb", _dp), ("bnd_ub", _dp),
        ("TolCon", C.c_double), ("WeightThreshold", C.c_double), ("WeightPenalty", C.c_double),
        ("sparse_cutoff", C.c_double),
        ("F", _dp), ("dF", _dp), ("G", _dp), ("H", _dp), ("dG", _dp), ("dH", _dp),
        ("varG", _dp), ("varGss", _dp), ("I_sk", _dp), ("J_sjk", _dp),
        ("G_s", _dp), ("varG_s", _dp),
        ("chunk_world", C.c_int32),
        ("restart_offset", C.c_int32), ("restart_stride", C.c_int32),
        ("no_jacobian", C.c_int32),
        ("dvarG", _dp),
        ("dG_s", _dp),
        ("dvarG_s", _dp),
        ("plan_restarts", C.c_int32),
    ]


_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH
        )
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.vbmc_abi_version.restype = C.c_int
    lib.vbmc_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    lib.vbmc_ctx_destroy.argtypes = [vp]
    lib.vbmc_ctx_destroy.restype = None
    lib.vbmc_last_error.argtypes = [vp]
    lib.vbmc_last_error.restype = C.c_char_p
    lib.vbmc_ctx_synchronize.argtypes = [vp]
    lib.vbmc_ctx_set_profiling.argtypes = [vp, C.c_int]
    lib.vbmc_ctx_last_kernel_ms.argtypes = [vp, _dp, _dp]
    lib.vbmc_gp_upload.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   _dp, _dp, _dp, _dp, _dp, C.POINTER(C.c_uint8), C.POINTER(vp)]
    lib.vbmc_gp_free.argtypes = [vp, vp]
    lib.vbmc_gp_free.restype = None
    lib.vbmc_elbo_batch.argtypes = [vp, vp, C.POINTER(ElboArgs)]
    lib.vbmc_elbo_abandon.argtypes = [vp, C.c_int]
    lib.vbmc_elbo_submit.argtypes = [vp, vp, C.POINTER(ElboArgs), C.c_int]
    lib.vbmc_elbo_collect.argtypes = [vp, C.POINTER(ElboArgs), C.c_int]
    lib.vbmc_adam_batch.argtypes = [vp, vp, C.POINTER(ElboArgs), C.c_double, C.c_int, C.c_double, C.c_double, C.c_double,
                                    _dp, _dp, C.POINTER(C.c_int32), _dp, _dp, _dp]
    lib.vbmc_elbo_shard_size.argtypes = [vp, vp, C.POINTER(ElboArgs), C.c_int, C.POINTER(C.c_size_t)]
    lib.vbmc_elbo_shard_begin.argtypes = [vp, vp, C.POINTER(ElboArgs), C.c_int, C.c_int, vp]
    lib.vbmc_elbo_shard_finish.argtypes = [vp, vp, C.POINTER(ElboArgs), C.c_int, vp]
    lib.vbmc_rng_dump.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, _dp]
    lib.vbmc_entropy_plan.argtypes = [C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 4
    lib.vbmc_entropy_plan.restype = C.c_int
    lib.vbmc_device_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.vbmc_device_free.argtypes = [vp, vp]
    lib.vbmc_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    lib.vbmc_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    i32p = C.POINTER(C.c_int32)
    u8p = C.POINTER(C.c_uint8)
    lib.vbmc_gp_set_noise.argtypes = [vp, vp, i32p, _dp]
    lib.vbmc_gp_post.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i32p, _dp, _dp, _dp, _dp,
                                 _dp, _dp, _dp, _dp, u8p, C.POINTER(vp)]
    lib.vbmc_gp_pred.argtypes = [vp, vp, C.c_int, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp]
    lib.vbmc_gp_rank1_solves.argtypes = [vp, vp, _dp, _dp, _dp, _dp]
    lib.vbmc_gp_rank1_update.argtypes = [vp, vp, _dp, C.c_double, _dp, _dp, _dp, _dp, _dp, C.POINTER(vp)]
    lib.vbmc_acq_eval.argtypes = [vp, vp, C.c_int, _dp, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_double, C.c_int, C.c_double,
                                  _dp, _dp, _dp, _dp, _dp, _dp]
    lib.vbmc_acq_is_create.argtypes = [vp, vp, C.c_int, _dp, C.c_int, _dp, _dp, _dp, C.POINTER(vp)]
    lib.vbmc_acq_is_free.argtypes = [vp, vp]
    lib.vbmc_acq_is_free.restype = None
    lib.vbmc_acq_iqr_eval.argtypes = [vp, vp, vp, C.c_int, _dp, _dp, _dp, _dp, C.c_int, C.c_double, _dp, _dp, _dp]
    lib.vbmc_gp_nlz.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i32p, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp]
    lib.vbmc_test_exp.argtypes = [vp, C.c_int, C.c_int, _dp, _dp]
    lib.vbmc_sq_dist.argtypes = [vp, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp]
    # the communicator inside the library (abi_comm.hip)
    vpp = C.POINTER(vp)
    lib.vbmc_comm_create_all.argtypes = [C.c_int, C.POINTER(C.c_int), vpp]
    lib.vbmc_comm_unique_id.argtypes = [C.c_char_p]
    lib.vbmc_comm_create_rank.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, vpp]
    lib.vbmc_comm_destroy.argtypes = [vp]
    lib.vbmc_comm_destroy.restype = None
    for name in ("vbmc_comm_size", "vbmc_comm_local", "vbmc_comm_rank"):
        getattr(lib, name).argtypes = [vp]
        getattr(lib, name).restype = C.c_int
    lib.vbmc_comm_ctx.argtypes = [vp, C.c_int]
    lib.vbmc_comm_ctx.restype = vp
    lib.vbmc_comm_last_error.argtypes = [vp]
    lib.vbmc_comm_last_error.restype = C.c_char_p
    lib.vbmc_allgather_f64.argtypes = [vp, vpp, vpp, C.c_size_t]
    lib.vbmc_allgather_host_f64.argtypes = [vp, _dp, _dp, C.c_size_t]
    lib.vbmc_gp_upload_all.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       _dp, _dp, _dp, _dp, _dp, C.POINTER(C.c_uint8), vpp]
    lib.vbmc_gp_free_all.argtypes = [vp, vpp]
    lib.vbmc_gp_free_all.restype = None
    lib.vbmc_elbo_batch_multi.argtypes = [vp, vpp, C.POINTER(ElboArgs)]
    lib.vbmc_elbo_multi_submit.argtypes = [vp, vpp, C.POINTER(ElboArgs), C.c_int]
    lib.vbmc_elbo_multi_collect.argtypes = [vp, C.POINTER(ElboArgs), C.c_int]
    for name in DECLARED_OPTIONAL:
        if hasattr(lib, name):
            pass
    if lib.vbmc_abi_version() != ABI_VERSION:
        raise ImportError("libvbmc_hip.so ABI version mismatch")
    _lib = lib
    return lib


DECLARED_OPTIONAL = ()


def f64(a, order="F"):
    """Contiguous column-major fp64 copy/view (what MATLAB hands over)."""
    return np.require(np.asarray(a, dtype=np.float64), dtype=np.float64, requirements=["F" if order == "F" else "C", "A"])


def ptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


class Context:
    """One HIP stream + device scratch on one GPU (vbmc_ctx)."""

    def __init__(self, device=0, stream=None):
        self.lib = load()
        h = C.c_void_p()
        st = self.lib.vbmc_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        if st != VBMC_OK:
            raise VbmcHipError(st, "vbmc_ctx_create(device=%d) failed: a gfx950 (MI355X) device is required; "
                                   "there is no CPU fallback" % device)
        self.h = h
        self.device = device

    def check(self, st):
        if st == VBMC_OK:
            return
        msg = self.lib.vbmc_last_error(self.h).decode("utf-8", "replace")
        if st == VBMC_ERR_UNSUPPORTED:
            raise VbmcUnsupported(st, msg)
        raise VbmcHipError(st, msg)

    def synchronize(self):
        self.check(self.lib.vbmc_ctx_synchronize(self.h))

    def set_profiling(self, on=True):
        """on = 2: the dominant kernel timed ALONE (the log joint is not forked beside it), see include/vbmc_hip.h"""
        self.check(self.lib.vbmc_ctx_set_profiling(self.h, 2 if (on is not True and on == 2) else (1 if on else 0)))

    def last_kernel_ms(self):
        a, b = C.c_double(), C.c_double()
        self.check(self.lib.vbmc_ctx_last_kernel_ms(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def rng_dump(self, D, K, R, Ns, seed):
        """eps (R, K, Ns/2, D) that eps_mode 0 consumes for `seed` (test hook)."""
        Mh = (Ns + 1) // 2
        out = np.empty((R, K, Mh, D), dtype=np.float64)
        self.check(self.lib.vbmc_rng_dump(self.h, D, K, R, Ns, C.c_uint64(seed), ptr(out)))
        return out

    def test_exp(self, x, variant=0):
        x = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))
        y = np.empty_like(x)
        self.check(self.lib.vbmc_test_exp(self.h, x.size, int(variant), ptr(x), ptr(y)))
        return y

    def close(self):
        if getattr(self, "h", None):
            self.lib.vbmc_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceGP:
    """Device-resident gp.post(1..S) (vbmc_gp)."""

    def __init__(self, ctx, X, hyp, alpha, L, sW1, Lchol, meanfun, Ncov, Nnoise):
        self.ctx = ctx
        X = f64(X)
        hyp = f64(hyp)
        if hyp.ndim == 1:
            hyp = hyp[:, None]
        alpha = f64(alpha)
        if alpha.ndim == 1:
            alpha = alpha[:, None]
        N, D = X.shape
        Nhyp, S = hyp.shape
        sW1 = f64(np.asarray(sW1).reshape(S))
        lch = np.ascontiguousarray(np.asarray(Lchol, dtype=np.uint8).reshape(S))
        Lp = None
        if L is not None:
            L = f64(L)  # (N, N, S) column-major == MATLAB N x N x S
            assert L.shape == (N, N, S), L.shape
            Lp = ptr(L)
        h = C.c_void_p()
        ctx.check(ctx.lib.vbmc_gp_upload(ctx.h, N, D, S, Nhyp, int(Ncov), int(Nnoise), int(meanfun), ptr(X), ptr(hyp),
                                         ptr(alpha), Lp, ptr(sW1), lch.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(h)))
        self.h = h
        self.N, self.D, self.S = N, D, S

    @classmethod
    def from_handle(cls, ctx, h, N, D, S):
        self = cls.__new__(cls)
        self.ctx, self.h, self.N, self.D, self.S = ctx, h, N, D, S
        return self

    def set_noise(self, noisefun, sn2_mult):
        nf = (C.c_int32 * 3)(*[int(x) for x in (list(noisefun) + [0, 0, 0])[:3]])
        m = f64(np.asarray(sn2_mult, dtype=np.float64).reshape(self.S))
        self.ctx.check(self.ctx.lib.vbmc_gp_set_noise(self.ctx.h, self.h, nf, ptr(m)))

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.vbmc_gp_free(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
