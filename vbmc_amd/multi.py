"""More than one GPU through the library's OWN communicator (include/vbmc_hip.h "communicator", vbmc_amd/csrc/abi_comm.hip):
RCCL is reached from inside libvbmc_hip.so, so this module needs no torch.distributed for the data path -- the same entry
points a MATLAB host binds through matlab/vbmc_hip_mex.cpp ('comm_open', 'elbo_batch' with a communicator).

    comm = Comm.create_all()                  # ONE process, every gfx950 device of the node (ncclCommInitAll)
    comm = Comm.from_torch(engine.ctx)        # one process PER device under torch.distributed.run: the 128-byte RCCL id
                                              # travels through the launcher's process group, the data path does not
    gps = comm.upload_gp(gp)                  # a replica of the surrogate on every local device
    out = comm.negelcbo_batch(thetas, 0, vp, gps, Ns, ...)    # the R restarts dealt r = g (mod G); F / varG all-gathered

    po = comm.prepare(T, R, 0, vp, gps, Ns)   # the same objective with everything resolved once; po(thetas, seed) blocks,
    po.submit(thetas, seed, slot); F, dF = po.collect(slot)   # ... or up to four batches in flight (vbmc_elbo_multi_submit / _collect)

The restart axis (misc/vpsieve_vbmc.m:74-78, misc/vpoptimize_vbmc.m:49) is the one the path shards over.  Every rank
evaluates the estimator the one-GPU batch evaluates, sample for sample (the device stream of a restart is keyed by its index in the
undivided batch); by default the values agree with the one-GPU batch to the order of summation (1e-13) -- bit for bit where the
per-device launch shapes coincide -- and all ranks of a call see the identical vectors; ``exact=True`` (round 5:
vbmc_elbo_args.plan_restarts) makes every device launch the undivided batch's shapes: bit-identical to one device, always."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import VBMC_ERR_UNSUPPORTED, VBMC_OK, VbmcHipError, VbmcUnsupported, f64, load, ptr
from .elbo import _build_args


class Comm:
    def __init__(self, handle):
        self.lib = load()
        self.h = handle
        self.size = self.lib.vbmc_comm_size(handle)
        self.local = self.lib.vbmc_comm_local(handle)
        self.rank = self.lib.vbmc_comm_rank(handle)

    # ---- construction
    @classmethod
    def create_all(cls, ndev=None, devices=None):
        lib = load()
        if devices is None:
            if ndev is None:
                import torch

                ndev = torch.cuda.device_count()
            devices = list(range(int(ndev)))
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        st = lib.vbmc_comm_create_all(len(devices), arr, C.byref(h))
        if st != VBMC_OK:
            raise VbmcHipError(st, "vbmc_comm_create_all(%r) failed (gfx950 devices and librccl are required)" % (devices,))
        return cls(h)

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        st = load().vbmc_comm_unique_id(buf)
        if st != VBMC_OK:
            raise VbmcHipError(st, "vbmc_comm_unique_id failed (librccl not loadable?)")
        return buf.raw

    @classmethod
    def create_rank(cls, ctx, rank, world, uid):
        assert len(uid) == 128
        h = C.c_void_p()
        st = ctx.lib.vbmc_comm_create_rank(ctx.h, int(rank), int(world), uid, C.byref(h))
        if st != VBMC_OK:
            ctx.check(st)
        c = cls(h)
        c._ctx = ctx          # the context stays the caller's: keep it alive as long as the communicator
        return c

    @classmethod
    def from_torch(cls, ctx, group=None):
        """One rank per process of an initialised torch.distributed group; only the 128-byte id uses that group."""
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls.create_rank(ctx, rank, world, box[0])

    # ---- plumbing
    def check(self, st):
        if st == VBMC_OK:
            return
        msg = self.lib.vbmc_comm_last_error(self.h).decode("utf-8", "replace")
        raise (VbmcUnsupported if st == VBMC_ERR_UNSUPPORTED else VbmcHipError)(st, msg)

    def close(self):
        if getattr(self, "h", None):
            self.lib.vbmc_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # ---- surrogate replicas
    def upload_gp(self, gp, need_L=False):
        post = gp["post"]
        S = len(post)
        X = f64(gp["X"])
        N, D = X.shape
        hyp = f64(np.stack([np.asarray(p["hyp"], dtype=np.float64).reshape(-1) for p in post], axis=1))
        alpha = f64(np.stack([np.asarray(p["alpha"], dtype=np.float64).reshape(-1) for p in post], axis=1))
        L = f64(np.stack([np.asarray(p["L"], dtype=np.float64) for p in post], axis=2)) if need_L else None
        sW1 = f64(np.array([np.asarray(p["sW"]).reshape(-1)[0] for p in post]))
        lch = np.ascontiguousarray([1 if p["Lchol"] else 0 for p in post], dtype=np.uint8)
        gps = (C.c_void_p * self.local)()
        self.check(self.lib.vbmc_gp_upload_all(self.h, N, D, S, hyp.shape[0], int(gp["Ncov"]), int(gp["Nnoise"]), int(gp["meanfun"]),
                                               ptr(X), ptr(hyp), ptr(alpha), ptr(L), ptr(sW1), lch.ctypes.data_as(C.POINTER(C.c_uint8)), gps))
        return gps

    def free_gp(self, gps):
        self.lib.vbmc_gp_free_all(self.h, gps)

    # ---- the exchange on its own
    def allgather_host(self, send):
        """send: (local, count) host doubles -> (size, count): every rank's block in rank order."""
        send = np.ascontiguousarray(np.asarray(send, dtype=np.float64).reshape(self.local, -1))
        recv = np.empty((self.size, send.shape[1]))
        self.check(self.lib.vbmc_allgather_host_f64(self.h, ptr(send), ptr(recv), send.shape[1]))
        return recv

    # ---- the batched objective dealt over the ranks
    def negelcbo_batch(self, thetas, beta, vp, gps, Ns=0, compute_grad=True, compute_var=None, thetabnd=None, *, separate_K=False,
                       seed=0, S=None, outputs=None, exact=False):
        """negelcbo_batch (vbmc_amd/elbo.py) for the UNDIVIDED batch thetas (T, R), identical on every rank: F and varG of all R
        restarts on every rank; the other outputs for the restarts this process evaluated (all of them with create_all), NaN
        elsewhere.  exact=True (vbmc_elbo_args.plan_restarts = R): every device launches the shapes of the undivided batch, so
        that the values are BIT-IDENTICAL to the one-device evaluation of the whole batch -- the sieve's order provably the same on
        1 and on N devices -- at the price of launch shapes chosen for R restarts on a device that holds R / G of them."""
        thetas = f64(thetas)
        if thetas.ndim == 1:
            thetas = f64(thetas.reshape(-1, 1))
        T, R = thetas.shape
        K = int(vp["K"])
        a, keep, compute_var = _build_args(thetas, beta, vp, None, Ns, compute_grad, compute_var, thetabnd, separate_K, None, None, False,
                                           seed, None)
        a.plan_restarts = R if exact else 0
        out = {}

        def buf(name, shape):
            if outputs is not None and name not in outputs:
                return None
            arr = np.full(shape, np.nan, dtype=np.float64, order="F")
            out[name] = arr
            return ptr(arr)

        a.F = buf("F", (R,)); a.G = buf("G", (R,)); a.H = buf("H", (R,))
        a.varG = buf("varG", (R,)); a.varGss = buf("varGss", (R,))
        if compute_grad:
            a.dF = buf("dF", (T, R)); a.dG = buf("dG", (T, R)); a.dH = buf("dH", (T, R))
        if separate_K:
            if S is None:
                raise ValueError("separate_K needs S = number of hyper-samples")
            a.I_sk = buf("I_sk", (S, K, R))
            if compute_var:
                a.J_sjk = buf("J_sjk", (S, K, K, R))
        self.check(self.lib.vbmc_elbo_batch_multi(self.h, gps, C.byref(a)))
        return out

    def prepare(self, T, R, beta, vp, gps, Ns, compute_var=0, thetabnd=None, exact=False):
        """The batched objective for a stream of batches of the same shape, resolved once (PreparedMulti)."""
        return PreparedMulti(self, T, R, beta, vp, gps, Ns, compute_var, thetabnd, exact)


class PreparedMulti:
    """negelcbo_vbmc(theta, beta, vp, gp, Ns, 1, compute_var, 0, thetabnd) for batches (T, R) dealt over the communicator's ranks
    (the multi-GPU sibling of vbmc_amd.elbo.PreparedObjective): the argument struct, the fixed vp groups, the bounds and the output
    buffers are built once; a call copies the new thetas into place.  __call__ blocks (vbmc_elbo_batch_multi); submit / collect keep
    up to four batches in flight (slots 0 .. 3; vbmc_elbo_multi_submit / vbmc_elbo_multi_collect): F of ALL R restarts, dF of the restarts this process
    evaluated (NaN elsewhere); with a variance term (compute_var != 0) also varG of ALL R restarts (``self.varG``; the third value
    ``collect`` returns) -- those passes stay on the contexts' own streams, slots 0 and 1 only (include/vbmc_hip.h).
    exact=True: vbmc_elbo_args.plan_restarts = R, bit-identical to the one-device batch (Comm.negelcbo_batch)."""

    def __init__(self, comm, T, R, beta, vp, gps, Ns, compute_var=0, thetabnd=None, exact=False):
        self.comm, self.gps = comm, gps
        self.theta = np.zeros((T, R), order="F")
        self.args, self._keep, cv = _build_args(self.theta, beta, vp, None, Ns, True, compute_var, thetabnd, False, None, None, False, 0, None)
        self.compute_var = int(cv)
        self.args.plan_restarts = R if exact else 0
        self._slots = {}
        self.F, self.dF, self.varG = self._buffers(self.args)

    def _buffers(self, a):
        T, R = self.theta.shape
        F = np.full(R, np.nan)
        dF = np.full((T, R), np.nan, order="F")
        a.F, a.dF = ptr(F), ptr(dF)
        varG = None
        if self.compute_var:           # (the library gathers [F | varG]; the buffer was never handed over)
            varG = np.full(R, np.nan)
            a.varG = ptr(varG)
        return F, dF, varG

    def _slot(self, slot):
        if self.compute_var and slot > 1:
            raise ValueError("passes with a variance term run on the contexts' own streams: slots 0 and 1 only")
        if slot not in self._slots:
            a = type(self.args).from_buffer_copy(self.args)
            F, dF, varG = self._buffers(a)
            self._slots[slot] = (a, F, dF, varG)
        return self._slots[slot]

    def __call__(self, thetas, seed=0):
        np.copyto(self.theta, np.asarray(thetas, dtype=np.float64).reshape(self.theta.shape, order="F"))
        self.args.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.comm.check(self.comm.lib.vbmc_elbo_batch_multi(self.comm.h, self.gps, C.byref(self.args)))
        return self.F, self.dF

    def submit(self, thetas, seed=0, slot=0):
        a = self._slot(slot)[0]
        np.copyto(self.theta, np.asarray(thetas, dtype=np.float64).reshape(self.theta.shape, order="F"))   # copied by the library before it returns
        a.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.comm.check(self.comm.lib.vbmc_elbo_multi_submit(self.comm.h, self.gps, C.byref(a), int(slot)))

    def collect(self, slot=0):
        a, F, dF, varG = self._slot(slot)
        self.comm.check(self.comm.lib.vbmc_elbo_multi_collect(self.comm.h, C.byref(a), int(slot)))
        return (F, dF, varG) if self.compute_var else (F, dF)
