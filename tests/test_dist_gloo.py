"""CPU, world_size 2, gloo: the restart-sharded sieve gathers to the same index-identical order
as the unsharded evaluation.  The device evaluation is replaced by the oracle (this test runs
where there is no GPU); what is under test is the sharding + all-gather + stable sort."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import vbmc_amd.optimize as opt
    from oracle import vbmc_ref as R
    from tests._cases import synth_problem
    from vbmc_amd import dist as vd

    p = synth_problem(3, 3, 25, 3, 2)
    gp = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=4)

    def fake_batch(thetas, beta, vp, gp_, Ns, compute_grad, compute_var, thetabnd, **kw):
        thetas = np.asarray(thetas)
        F = np.array([R.negelcbo_vbmc(thetas[:, r], 0, vp, gp_, 0, False, 0, thetabnd=thetabnd)["F"] for r in range(thetas.shape[1])])
        return {"F": F, "varG": np.zeros_like(F)}

    opt.negelcbo_batch = fake_batch  # stand-in for the HIP batch on a GPU-less host
    vp = R.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    kw = dict(rng=np.random.default_rng(5))
    single = opt.vpsieve_vbmc(11, 3, vp, gp, **kw)
    kw = dict(rng=np.random.default_rng(5))
    sharded = opt.vpsieve_vbmc(11, 3, vp, gp, shard=vd.shard_spec(), **kw)
    ok = np.array_equal(single[6], sharded[6]) and np.array_equal(single[1], sharded[1])
    ok = ok and all(np.array_equal(a["mu"], b["mu"]) for a, b in zip(single[0], sharded[0]))
    q.put((rank, bool(ok), sharded[6].tolist()))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_sharded_sieve_is_index_identical():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=150) for _ in procs]
    for pr in procs:
        pr.join(timeout=30)
    assert all(r[1] for r in res), res
    assert res[0][2] == res[1][2]  # both ranks hold the identical ELCBO vector
    assert not np.any(np.isnan(res[0][2]))


def _acq_worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import vbmc_amd.acq as acq
    from oracle import vbmc_ref as R
    from tests._cases import synth_problem
    from vbmc_amd import dist as vd

    p = synth_problem(4, 3, 25, 3, 2)
    gp = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=4)
    vp = R.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))

    def fake_local(Xs, vp_, gp_, st, acqFun, outside, nargout, engine, transpose_flag=False):   # oracle stands in for the device
        a, fb, vt = R.acqwrapper_vbmc(Xs, vp_, gp_, st, acqFun.replace("_vbmc", ""), outside)
        return (a, fb, vt) if nargout >= 3 else a

    acq._acq_local = fake_local
    Xs = np.random.default_rng(2).standard_normal((37, 3))            # 37 points: uneven shards
    outside = np.zeros(37, dtype=bool)
    outside[[3, 20]] = True
    st = {"ymax": float(np.max(p["y"])), "VarianceRegularizedAcqFcn": True, "TolGPVar": 1e-4}
    single = acq.acqwrapper_vbmc(Xs, vp, gp, st, False, "acqf_vbmc", None, outside=outside)
    sharded, fb, vt = acq.acqwrapper_vbmc(Xs, vp, gp, st, False, "acqf_vbmc", None, outside=outside, nargout=3, shard=vd.shard_spec())
    # values agree to rounding (sq_dist centres on the mean of the points it is given, hence per shard); the argmin is identical
    fin = np.isfinite(single)
    ok = np.array_equal(fin, np.isfinite(sharded)) and np.allclose(single[fin], sharded[fin], rtol=1e-11, atol=0)
    ok = ok and int(np.argmin(single)) == int(np.argmin(sharded)) and fb.shape == (37,)
    q.put((rank, bool(ok), int(np.argmin(sharded))))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_sharded_acquisition_sweep_picks_the_same_point():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_acq_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=150) for _ in procs]
    for pr in procs:
        pr.join(timeout=30)
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2]


def _exchange_worker(rank, world, port, q):
    import ctypes

    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vbmc_amd.dist import ShardExchange

    ex = ShardExchange(device=torch.device("cpu"))      # the exchange of negelcbo_shard, on host tensors
    ok = True
    for n in (7, 1000, 7):                                # buffers are re-made when the block size changes
        p = ex.send_buffer(n)
        mine = (np.arange(n, dtype=np.float64) + 1000.0 * rank)
        ctypes.memmove(p, mine.ctypes.data, 8 * n)        # what vbmc_elbo_shard_begin does on the device
        g = ex.all_gather()
        got = np.ctypeslib.as_array(ctypes.cast(g, ctypes.POINTER(ctypes.c_double)), shape=(world * n,)).copy()
        want = np.concatenate([np.arange(n, dtype=np.float64) + 1000.0 * r for r in range(world)])   # rank order
        ok = ok and np.array_equal(got, want)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_shard_exchange_gathers_the_blocks_in_rank_order():
    """The one collective of the hyper-sample / sample-chunk sharded evaluation (vbmc_amd.dist.ShardExchange): every rank ends
    with all blocks in rank order -- what vbmc_elbo_shard_finish expects.  (The bit-identity of the sharded evaluation itself
    needs the device: tests/test_gpu_shard_s.py.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=150) for _ in procs]
    for pr in procs:
        pr.join(timeout=30)
    assert all(ok for _, ok in res), res
