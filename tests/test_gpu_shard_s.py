"""GPU: ONE evaluation sharded along the GP hyper-sample axis and the entropy's sample chunks (SURVEY 8e, R < G) is
BIT-IDENTICAL to the unsharded evaluation with the same chunking (chunk_world = world) -- same summation order
(vbmc_elbo_shard_begin / _finish).
World sizes up to 8 are emulated in one process on one GPU (each rank's block is computed in turn and the blocks are
concatenated as an all-gather would); a real two-process run follows."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests._cases import synth_problem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class LocalExchange:
    """Stands in for the all-gather of vbmc_amd.dist.ShardExchange inside one process: rank g's block is written at offset g n
    of one device buffer from the library's own allocator (vbmc_device_alloc: no torch in this process)."""

    def __init__(self, ctx, world):
        self.ctx, self.world, self.ptr, self.n, self.rank = ctx, world, None, 0, 0

    def send_buffer(self, n):
        import ctypes as C

        if self.ptr is None or self.n != n:
            self.free()
            p = C.c_void_p()
            self.ctx.check(self.ctx.lib.vbmc_device_alloc(self.ctx.h, C.c_size_t(8 * n * self.world), C.byref(p)))
            self.ptr, self.n = p.value, n
        return self.ptr + 8 * n * self.rank

    def all_gather(self):
        return self.ptr

    def free(self):
        import ctypes as C

        if self.ptr is not None:
            self.ctx.lib.vbmc_device_free(self.ctx.h, C.c_void_p(self.ptr))
            self.ptr = None


def _setup(va, seed, D, N, K, S):
    from oracle import vbmc_ref as R

    p = synth_problem(seed, D, N, K, S)
    gp = va.gplite_post(p["hyp"], p["X"], p["y"], 1, 4, (1, 0, 0), None)
    vp = va.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    theta = np.concatenate([p["mu"].reshape(-1, order="F"), np.log(p["sigma"]), np.log(p["lam"]), p["eta"]])
    _, tb = R.vpbounds(vp, gp, dict(TolLength=1e-6, TolWeight=1e-2, TolConLoss=0.01, WeightPenalty=0.1))
    return gp, vp, theta, tb


def _sharded(va, world, thetas, vp, gp, Ns, tb, seed, grad=True):
    import ctypes as C

    from vbmc_amd import elbo as E

    eng = va.default_engine()
    ctx = eng.ctx
    ex = LocalExchange(ctx, world)
    th = E.f64(thetas if thetas.ndim == 2 else thetas.reshape(-1, 1))
    a, keep, _ = E._build_args(th, 0.0, vp, gp, Ns, grad, 0, tb, False, None, None, False, seed, eng)
    dgp = eng.device_gp(gp)
    T, R = th.shape
    out = {"F": np.empty(R), "G": np.empty(R), "H": np.empty(R)}
    if grad:
        out.update(dF=np.empty((T, R), order="F"), dG=np.empty((T, R), order="F"), dH=np.empty((T, R), order="F"))
    for k, v in out.items():
        setattr(a, k, E.ptr(v))
    n = C.c_size_t(0)
    ctx.check(ctx.lib.vbmc_elbo_shard_size(ctx.h, dgp.h, C.byref(a), world, C.byref(n)))
    for g in range(world):
        ex.rank = g
        ctx.check(ctx.lib.vbmc_elbo_shard_begin(ctx.h, dgp.h, C.byref(a), g, world, C.c_void_p(ex.send_buffer(n.value))))
    ctx.check(ctx.lib.vbmc_elbo_shard_finish(ctx.h, dgp.h, C.byref(a), world, C.c_void_p(ex.all_gather())))
    ex.free()
    return out


@pytest.mark.parametrize("cfg", [(10, 400, 50, 20, 10000, 1), (4, 60, 5, 3, 200, 1), (6, 100, 10, 8, 1000, 2), (5, 80, 70, 5, 600, 1),
                                 (3, 40, 4, 2, 0, 1), (10, 200, 50, 20, 10000, 3)])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_evaluation_is_bit_identical(cfg, world):
    import vbmc_amd as va

    D, N, K, S, Ns, R = cfg
    gp, vp, theta, tb = _setup(va, 11, D, N, K, S)
    rng = np.random.default_rng(1)
    thetas = np.asfortranarray(theta[:, None] + 0.05 * rng.standard_normal((theta.size, R)))
    ref = va.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, tb, seed=99, chunk_world=world)
    plain = va.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, tb, seed=99)   # default chunking: same numbers to summation order
    assert np.allclose(plain["F"], ref["F"], rtol=1e-12, atol=0) and np.allclose(plain["dF"], ref["dF"], rtol=1e-9, atol=1e-12)
    got = _sharded(va, world, thetas, vp, gp, Ns, tb, 99)
    for k in ("F", "G", "H", "dF", "dG", "dH"):
        assert np.array_equal(got[k], ref[k]), (k, float(np.max(np.abs(got[k] - ref[k]))))
    ref0 = va.negelcbo_batch(thetas, 0, vp, gp, Ns, False, 0, tb, seed=99, chunk_world=world)
    got0 = _sharded(va, world, thetas, vp, gp, Ns, tb, 99, grad=False)
    assert np.array_equal(got0["F"], ref0["F"]) and np.array_equal(got0["H"], ref0["H"])


def test_unsupported_forms_are_refused():
    import ctypes as C

    import vbmc_amd as va
    from vbmc_amd import elbo as E

    gp, vp, theta, tb = _setup(va, 12, 3, 30, 3, 2)
    eng = va.default_engine()
    a, keep, _ = E._build_args(E.f64(theta.reshape(-1, 1)), 0.0, vp, gp, 100, False, 1, None, False, None, None, False, 0, eng)
    n = C.c_size_t(0)
    st = eng.ctx.lib.vbmc_elbo_shard_size(eng.ctx.h, eng.device_gp(gp, need_L=True).h, C.byref(a), 2, C.byref(n))
    assert st == 4   # VBMC_ERR_UNSUPPORTED: variance is outside the sharded form


_WORKER = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
nd = torch.cuda.device_count()
torch.cuda.set_device(rank %% nd)
backend = "nccl" if nd >= world else "gloo"
dist.init_process_group(backend, **({"device_id": torch.device("cuda", rank %% nd)} if backend == "nccl" else {}))
import vbmc_amd as va
from vbmc_amd.dist import ShardExchange
from tests.test_gpu_shard_s import _setup
eng = va.Engine(rank %% nd)
gp, vp, theta, tb = _setup(va, 11, 10, 400, 50, 20)
gp = va.gplite_post(np.stack([p["hyp"] for p in gp["post"]], axis=1), gp["X"], gp["y"], 1, 4, (1, 0, 0), None, engine=eng)
ex = ShardExchange(device=torch.device("cuda", rank %% nd))
ref = va.negelcbo_batch(theta, 0, vp, gp, 10000, True, 0, tb, seed=5, engine=eng, chunk_world=world)
got = va.negelcbo_shard(theta, 0, vp, gp, 10000, True, tb, rank=rank, world=world, exchange=ex, seed=5, engine=eng, outputs=("F", "dF", "H"))
ok = bool(np.array_equal(got["F"], ref["F"]) and np.array_equal(got["dF"], ref["dF"]) and np.array_equal(got["H"], ref["H"]))
allok = torch.tensor([1 if ok else 0], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
dist.all_reduce(allok, op=dist.ReduceOp.MIN)
if rank == 0:
    print(json.dumps({"ok": bool(allok.item()), "backend": backend, "world": world, "F": float(got["F"][0])}), flush=True)
dist.destroy_process_group()
"""


@pytest.mark.timeout(600)
def test_two_process_sharded_evaluation(tmp_path):
    """Two real ranks (RCCL when the box has two GPUs, otherwise gloo with both ranks on device 0): every rank ends with the
    bits of the unsharded evaluation."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT})
    port = 29500 + os.getpid() % 400
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=500) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1])
    assert line["ok"] and line["world"] == 2
