"""GPU: the communicator inside the library (vbmc_comm_*, vbmc_elbo_batch_multi: RCCL reached from libvbmc_hip.so itself).

On the one-GPU box: a one-rank communicator (ncclCommInitAll / ncclCommInitRank with world 1, a real ncclAllGather on the
context's stream) must return the bits of vbmc_elbo_batch; and the piece that makes a DEALT batch equal the undivided one --
the device stream of restart r keyed by its index in the whole batch (restart_offset / restart_stride) -- is checked by
evaluating the shares of G = 2, 3, 8 emulated ranks one after the other on the same device.  With two or more devices the same
test runs the real thing (skipped otherwise)."""
import ctypes as C

import numpy as np
import pytest

from oracle import vbmc_ref as R
from tests._cases import synth_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def va():
    import vbmc_amd

    return vbmc_amd


def setup(seed, D, N, K, S, Rn):
    p = synth_problem(seed, D, N, K, S)
    gp = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=p["meanfun"])
    vp = R.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    theta = np.concatenate([p["mu"].reshape(-1, order="F"), np.log(p["sigma"]), np.log(p["lam"]), p["eta"]])
    Th = np.asfortranarray(theta[:, None] + 0.05 * np.random.default_rng(seed).standard_normal((theta.size, Rn)))
    return gp, vp, Th


def test_dealt_shares_equal_the_undivided_batch(va):
    """Rank g's share (restarts g, g + G, ...) evaluated as its own batch with restart_offset = g, restart_stride = G gives,
    column for column, the bits of the undivided batch -- Monte-Carlo entropy on the device stream included."""
    from vbmc_amd.elbo import _build_args, default_engine
    from vbmc_amd._lib import ptr

    gp, vp, Th = setup(11, 5, 40, 7, 3, 13)
    eng = default_engine()
    whole = va.negelcbo_batch(Th, 0, vp, gp, 50, True, 0, seed=21)
    for G in (2, 3, 8):
        for g in range(G):
            cols = np.arange(Th.shape[1])[g::G]
            sub = np.asfortranarray(Th[:, cols])
            a, keep, _ = _build_args(sub, 0, vp, gp, 50, True, 0, None, False, None, None, False, 21, eng)
            a.restart_offset, a.restart_stride = g, G
            F = np.empty(cols.size); dF = np.empty((Th.shape[0], cols.size), order="F"); H = np.empty(cols.size)
            a.F, a.dF, a.H = ptr(F), ptr(dF), ptr(H)
            dgp = eng.device_gp(gp)
            eng.ctx.check(eng.ctx.lib.vbmc_elbo_batch(eng.ctx.h, dgp.h, C.byref(a)))
            assert np.array_equal(F, whole["F"][cols]) and np.array_equal(H, whole["H"][cols])
            assert np.array_equal(dF, whole["dF"][:, cols])
    # and the default (0 / 0) is the plain batch
    a, keep, _ = _build_args(Th, 0, vp, gp, 50, True, 0, None, False, None, None, False, 21, eng)   # (the same call form: the value-only
    # evaluation adds its log-joint records in another order -- separate kernel, no role in the entropy launch -- and agrees to 1e-15)
    F = np.empty(Th.shape[1]); a.F = ptr(F)
    dF = np.empty(Th.shape, order="F"); a.dF = ptr(dF)
    eng.ctx.check(eng.ctx.lib.vbmc_elbo_batch(eng.ctx.h, eng.device_gp(gp).h, C.byref(a)))
    assert np.array_equal(F, whole["F"])


def _check_multi(va, comm, with_var):
    gp, vp, Th = setup(12, 4, 35, 6, 2, 9)
    gps = comm.upload_gp(gp, need_L=with_var)
    try:
        for Ns, grad, cv in ((0, False, 1 if with_var else 0), (40, True, 0), (40, False, 1 if with_var else 0)):
            ref = va.negelcbo_batch(Th, 0, vp, gp, Ns, grad, cv, seed=5)
            out = comm.negelcbo_batch(Th, 0, vp, gps, Ns, grad, cv, seed=5)
            assert np.array_equal(out["F"], ref["F"]), (Ns, grad, cv)
            assert np.array_equal(out["varG"], ref["varG"])
            mine = np.arange(Th.shape[1])                    # create_all: every restart is local
            assert np.array_equal(out["G"][mine], ref["G"]) and np.array_equal(out["H"][mine], ref["H"])
            if grad:
                assert np.array_equal(out["dF"], ref["dF"]) and np.array_equal(out["dH"], ref["dH"])
        if with_var:
            ref = va.negelcbo_batch(Th, 0, vp, gp, 30, False, 1, separate_K=True, seed=6)
            out = comm.negelcbo_batch(Th, 0, vp, gps, 30, False, 1, separate_K=True, seed=6, S=2)
            assert np.array_equal(out["I_sk"], ref["I_sk"]) and np.array_equal(out["J_sjk"], ref["J_sjk"])
    finally:
        comm.free_gp(gps)


def test_one_rank_communicator_is_bit_identical(va):
    """ncclCommInitAll over ONE device: the whole vbmc_elbo_batch_multi path (deal, pick kernel, ncclAllGather inside a group
    on the context's stream, gathered read-back) against vbmc_elbo_batch."""
    from vbmc_amd.multi import Comm

    comm = Comm.create_all(1)
    assert (comm.size, comm.local, comm.rank) == (1, 1, 0)
    v = np.arange(5.0)
    assert np.array_equal(comm.allgather_host(v), v.reshape(1, 5))
    _check_multi(va, comm, with_var=True)
    # more ranks than restarts is fine too (R = 1 here: nothing to deal); refusals name their reason
    gp, vp, Th = setup(13, 3, 30, 4, 2, 1)
    gps = comm.upload_gp(gp)
    out = comm.negelcbo_batch(Th, 0, vp, gps, 20, True, 0, seed=3)
    assert np.array_equal(out["F"], va.negelcbo_batch(Th, 0, vp, gp, 20, True, 0, seed=3)["F"])
    comm.free_gp(gps)
    comm.close()


def test_pipelined_multi_equals_the_blocking_call(va):
    """vbmc_elbo_multi_submit / vbmc_elbo_multi_collect (Comm.prepare): two batches in flight through a one-rank communicator give,
    batch for batch, the bits of the blocking vbmc_elbo_batch_multi and of the one-device vbmc_elbo_batch; a second submit into a busy
    slot and a collect of an empty one are refused by name."""
    from vbmc_amd.multi import Comm
    from vbmc_amd._lib import VbmcHipError

    comm = Comm.create_all(1)
    gp, vp, Th = setup(14, 5, 40, 6, 3, 10)
    gps = comm.upload_gp(gp)
    T, Rn = Th.shape
    po = comm.prepare(T, Rn, 0, vp, gps, 48)
    batches = [np.asfortranarray(Th + 0.01 * i) for i in range(5)]
    refs = [va.negelcbo_batch(b, 0, vp, gp, 48, True, 0, seed=100 + i) for i, b in enumerate(batches)]
    got, pend = [], []
    for i, b in enumerate(batches):
        po.submit(b, seed=100 + i, slot=i & 1)
        pend.append(i & 1)
        if len(pend) == 2:
            F, dF = po.collect(pend.pop(0))
            got.append((F.copy(), dF.copy()))
    while pend:
        F, dF = po.collect(pend.pop(0))
        got.append((F.copy(), dF.copy()))
    for (F, dF), ref in zip(got, refs):
        assert np.array_equal(F, ref["F"]) and np.array_equal(dF, ref["dF"])
    F, dF = po(batches[2], seed=102)                       # the blocking form of the same prepared objective
    assert np.array_equal(F, refs[2]["F"]) and np.array_equal(dF, refs[2]["dF"])
    po.submit(batches[0], seed=1, slot=0)
    with pytest.raises(VbmcHipError, match="holds an uncollected"):
        po.submit(batches[1], seed=2, slot=0)
    po.collect(0)
    with pytest.raises(VbmcHipError, match="nothing submitted"):
        po.collect(1)
    comm.free_gp(gps)
    comm.close()


def test_dealt_shares_at_a_shape_where_the_launches_differ(va):
    """A batch large enough that the share of one of G = 8 ranks is launched differently from the undivided batch (the number of sample
    chunks per component follows the restarts a device holds: strong scaling wants that).  Every rank still evaluates the estimator of
    the undivided batch sample for sample, so the values agree to the order of summation -- 1e-12 here, against the 1e-6 the path
    promises -- and the sieve order of well-separated candidates is the same."""
    from vbmc_amd.elbo import _build_args, default_engine
    from vbmc_amd._lib import ptr

    gp, vp, Th = setup(15, 10, 60, 50, 4, 64)
    eng = default_engine()
    Ns = 2000
    whole = va.negelcbo_batch(Th, 0, vp, gp, Ns, True, 0, seed=31)
    G = 8
    F_all = np.empty(Th.shape[1])
    worst = 0.0
    for g in range(G):
        cols = np.arange(Th.shape[1])[g::G]
        sub = np.asfortranarray(Th[:, cols])
        a, keep, _ = _build_args(sub, 0, vp, gp, Ns, True, 0, None, False, None, None, False, 31, eng)
        a.restart_offset, a.restart_stride = g, G
        F = np.empty(cols.size); dF = np.empty((Th.shape[0], cols.size), order="F")
        a.F, a.dF = ptr(F), ptr(dF)
        eng.ctx.check(eng.ctx.lib.vbmc_elbo_batch(eng.ctx.h, eng.device_gp(gp).h, C.byref(a)))
        F_all[cols] = F
        worst = max(worst, float(np.max(np.abs(F - whole["F"][cols]) / np.maximum(1.0, np.abs(whole["F"][cols])))),
                    float(np.max(np.abs(dF - whole["dF"][:, cols])) / max(1.0, float(np.max(np.abs(whole["dF"]))))))
    assert worst < 1e-12, worst
    assert np.array_equal(np.argsort(F_all, kind="stable"), np.argsort(whole["F"], kind="stable"))
    assert worst > 0.0        # (the shape was chosen for this: here the two launches DO differ -- the exact mode below is not vacuous)
    # EXACT MODE (round 5, vbmc_elbo_args.plan_restarts = the undivided R): every share launches the undivided batch's shapes -- sample
    # chunks, log-joint kernel and its splits -- so every column is bit-identical, for shares of 2, 3 and 8 ranks (64 % 3 != 0: a ragged deal).
    # (Round 6: the undivided batch itself is evaluated under the same plan_restarts -- a blocking call without it may take the walking
    # entropy launch, whose records are summed in another order: 1e-13, not the same bits.)
    plain = whole
    whole = va.negelcbo_batch(Th, 0, vp, gp, Ns, True, 0, seed=31, plan_restarts=Th.shape[1])
    assert float(np.max(np.abs(whole["F"] - plain["F"]) / np.maximum(1.0, np.abs(plain["F"])))) < 1e-12
    for G2 in (8, 3, 2):
        for g in range(G2):
            cols = np.arange(Th.shape[1])[g::G2]
            r = va.negelcbo_batch(np.asfortranarray(Th[:, cols]), 0, vp, gp, Ns, True, 0, seed=31, restart_offset=g, restart_stride=G2,
                                  plan_restarts=Th.shape[1], outputs=("F", "dF", "H", "G"))
            assert np.array_equal(r["F"], whole["F"][cols]) and np.array_equal(r["dF"], whole["dF"][:, cols])
            assert np.array_equal(r["H"], whole["H"][cols]) and np.array_equal(r["G"], whole["G"][cols])
    # plan_restarts below the share's own size is refused
    with pytest.raises(va.VbmcHipError, match="plan_restarts"):
        va.negelcbo_batch(Th, 0, vp, gp, Ns, True, 0, seed=31, plan_restarts=5)


def test_exact_mode_through_the_communicator(va):
    """The same through vbmc_elbo_batch_multi / vbmc_elbo_multi_submit with a one-rank communicator: exact = True passes plan_restarts
    through and returns the plain batch's bits (one rank: trivially the same launch), and a variance pass hands varG back (ADVICE r4:
    PreparedMulti accepted compute_var and never returned the gathered varG)."""
    from vbmc_amd.multi import Comm

    gp, vp, Th = setup(16, 6, 50, 12, 3, 16)
    comm = Comm.create_all(1)
    gps = comm.upload_gp(gp, need_L=True)
    ref = va.negelcbo_batch(Th, 0, vp, gp, 400, True, 0, seed=9, outputs=("F", "dF"))
    out = comm.negelcbo_batch(Th, 0, vp, gps, 400, True, 0, seed=9, outputs=("F", "dF"), exact=True)
    assert np.array_equal(out["F"], ref["F"]) and np.array_equal(out["dF"], ref["dF"])
    po = comm.prepare(Th.shape[0], Th.shape[1], 0, vp, gps, 400, exact=True)
    po.submit(Th, seed=9, slot=2)
    F, dF = po.collect(2)
    assert np.array_equal(F, ref["F"]) and np.array_equal(dF, ref["dF"])
    refv = va.negelcbo_batch(Th, 1.0, vp, gp, 0, True, 2, outputs=("F", "dF", "varG"))
    pv = comm.prepare(Th.shape[0], Th.shape[1], 1.0, vp, gps, 0, compute_var=2)
    pv.submit(Th, seed=0, slot=1)
    Fv, dFv, varGv = pv.collect(1)
    assert np.array_equal(Fv, refv["F"]) and np.array_equal(varGv, refv["varG"]) and np.array_equal(dFv, refv["dF"])
    with pytest.raises(ValueError, match="slots 0 and 1"):
        pv.submit(Th, seed=0, slot=2)
    comm.free_gp(gps)
    comm.close()


def test_two_ranks_on_this_box(va):
    """more than ONE rank through the library's multi-device path on a one-GPU box.  RCCL itself cannot do it: with
    the device listed twice ncclCommInitAll -- and, across two processes, ncclCommInitRank -- answers "invalid usage" ("Duplicate GPU
    detected : rank 0 and rank 1 both on CUDA device": measured on the box in round 5, RCCL 2.27.7; there is no switch).  What CAN run
    here is everything around the collective with two local ranks: vbmc_comm_create_all(2, {0, 0}) gives two contexts (own streams, own
    slots) on device 0 and exchanges by event-ordered device-to-device copies instead of ncclAllGather (abi_comm.hip: local_copy) -- the
    deal r = g (mod 2) of a RAGGED batch (R = 7: 4 + 3), k_comm_pick's NaN padding of the shorter share, the per-device argument
    structs, vbmc_elbo_batch_multi and four vbmc_elbo_multi_submit / _collect batches in flight over two local devices, and the host
    all-gather.  Every value must equal the one-device batch bit for bit (exact mode), on every slot, in any collect order."""
    from vbmc_amd.multi import Comm

    gp, vp, Th = setup(18, 5, 40, 9, 3, 7)
    ref = va.negelcbo_batch(Th, 0, vp, gp, 120, True, 0, seed=4, outputs=("F", "dF"))
    comm = Comm.create_all(devices=[0, 0])
    assert comm.size == 2 and comm.local == 2
    gps = comm.upload_gp(gp)
    out = comm.negelcbo_batch(Th, 0, vp, gps, 120, True, 0, seed=4, outputs=("F", "dF"), exact=True)
    assert np.array_equal(out["F"], ref["F"]) and np.array_equal(out["dF"], ref["dF"])
    fast = comm.negelcbo_batch(Th, 0, vp, gps, 120, True, 0, seed=4, outputs=("F", "dF"))          # default mode: to the order of summation
    assert np.max(np.abs(fast["F"] - ref["F"]) / np.abs(ref["F"])) < 1e-12
    po = comm.prepare(Th.shape[0], Th.shape[1], 0, vp, gps, 120, exact=True)
    for sl in (0, 1, 2, 3):
        po.submit(Th, seed=4, slot=sl)
    for sl in (3, 1, 0, 2):
        F, dF = po.collect(sl)
        assert np.array_equal(F, ref["F"]) and np.array_equal(dF, ref["dF"])
    # three ranks, R = 7 again (3 + 2 + 2), and a batch smaller than the world (R = 2 on three ranks: one rank holds nothing)
    comm3 = Comm.create_all(devices=[0, 0, 0])
    gps3 = comm3.upload_gp(gp)
    out3 = comm3.negelcbo_batch(Th, 0, vp, gps3, 120, True, 0, seed=4, outputs=("F", "dF"), exact=True)
    assert np.array_equal(out3["F"], ref["F"]) and np.array_equal(out3["dF"], ref["dF"])
    ref2 = va.negelcbo_batch(np.asfortranarray(Th[:, :2]), 0, vp, gp, 120, True, 0, seed=4, outputs=("F", "dF"))
    out2 = comm3.negelcbo_batch(np.asfortranarray(Th[:, :2]), 0, vp, gps3, 120, True, 0, seed=4, outputs=("F", "dF"), exact=True)
    assert np.array_equal(out2["F"], ref2["F"]) and np.array_equal(out2["dF"], ref2["dF"])
    blocks = np.arange(9.0).reshape(3, 3)
    assert np.array_equal(comm3.allgather_host(blocks), blocks)
    comm3.free_gp(gps3); comm3.close()
    comm.free_gp(gps); comm.close()


# ---- restored in round 6 (deleted in round 5 with no replacement): the RCCL-missing error path, the rank form bench.py
# uses (ncclGetUniqueId + vbmc_comm_create_rank), all the devices of a node from one process, and bench.py's own N > 1 code path

def test_a_host_without_rccl_reports_instead_of_crashing():
    """with librccl unloadable every communicator entry point must return VBMC_ERR_HIP with a message (the first version built
    the message from two dlerror() calls, the second of which returns NULL).  A fresh process, RCCL disabled by the test hook."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from vbmc_amd.multi import Comm\n"
            "from vbmc_amd._lib import VbmcHipError\n"
            "for f in (lambda: Comm.create_all(1), Comm.unique_id):\n"
            "    try:\n        f(); print('NO ERROR')\n    except VbmcHipError as e:\n        print('refused:', e)\n") % root
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VBMC_RCCL_DISABLE="1"), capture_output=True, text=True, cwd=root, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-1500:])
    assert r.stdout.count("refused:") == 2 and "NO ERROR" not in r.stdout, r.stdout


def test_rank_form_with_a_unique_id(va):
    """ncclGetUniqueId + ncclCommInitRank (the one-process-per-GPU form bench.py uses), world 1 on this box, on the default
    engine's own context."""
    from vbmc_amd.multi import Comm

    ctx = va.default_engine().ctx
    comm = Comm.create_rank(ctx, 0, 1, Comm.unique_id())
    assert comm.size == 1 and comm.local == 1
    _check_multi(va, comm, with_var=False)
    r = comm.allgather_host(np.array([1.5, -2.0]))
    assert r.shape == (1, 2) and r[0, 1] == -2.0
    comm.close()


def test_all_devices_of_the_node(va):
    """With >= 2 gfx950 devices: ONE process drives them all; every value equals the one-device batch."""
    import torch
    from vbmc_amd.multi import Comm

    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("one device on this box: the multi-device form is exercised with emulated ranks above")
    comm = Comm.create_all(n)
    assert comm.size == n and comm.local == n
    _check_multi(va, comm, with_var=True)
    blocks = np.arange(3.0 * n).reshape(n, 3)
    assert np.array_equal(comm.allgather_host(blocks), blocks)
    comm.close()


def test_bench_multi_gpu_code_path_with_one_rank(va):
    """bench.py's N > 1 path -- process group over RCCL, the 128-byte id broadcast through it (Comm.from_torch), the surrogate uploaded
    through the communicator, vbmc_elbo_batch_multi + ncclAllGather inside the library every step, the strong-scaling leg, the timing
    rows gathered at the end -- executed under torch.distributed.run with ONE rank (VBMC_BENCH_FORCE_COMM=1): all a one-GPU box can
    run of what the driver launches on eight."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VBMC_BENCH_FORCE_COMM="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-aux",
                        "--no-cpu-baseline", "--restarts", "8"], capture_output=True, text=True, cwd=root, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["world_size_observed"] == 1
    assert "ncclAllGather inside libvbmc_hip.so" in d["exchange"], d["exchange"]
    assert d["strong"]["restarts_total"] == 8 and d["strong"]["value"] > 0
