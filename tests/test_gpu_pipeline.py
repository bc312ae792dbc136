"""vbmc_elbo_submit / vbmc_elbo_collect (include/vbmc_hip.h): the pipelined form of the batched evaluation for streams of
independent batches (the sieve's candidates, misc/vpsieve_vbmc.m:74-78).  Bit-identical to vbmc_elbo_batch, in order, with
other entry points of the same context called in between; misuse is refused."""
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


def setup(va, seed, D, N, K, S, Rr):
    p = synth_problem(seed, D, N, K, S)
    gp = va.gplite_post(p["hyp"], p["X"], p["y"], 1, p["meanfun"])
    vp = va.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    theta = np.concatenate([p["mu"].reshape(-1, order="F"), np.log(p["sigma"]), np.log(p["lam"]), p["eta"]])
    rng = np.random.default_rng(seed)
    batches = [np.asfortranarray(theta[:, None] + 0.05 * rng.standard_normal((theta.size, Rr))) for _ in range(5)]
    return p, gp, vp, batches


@pytest.mark.parametrize("cfg", [(4, 40, 5, 3, 8, 200), (10, 120, 50, 4, 16, 1000), (3, 30, 70, 2, 4, 64), (5, 50, 6, 2, 3, 0)])
def test_pipelined_batches_are_bit_identical_and_in_order(va, cfg):
    D, N, K, S, Rr, Ns = cfg
    p, gp, vp, batches = setup(va, 3, D, N, K, S, Rr)
    T = batches[0].shape[0]
    obj = va.PreparedObjective(T, Rr, 0, vp, gp, Ns, 0, None)
    ref = []
    for i, th in enumerate(batches):
        F, dF = obj(th, seed=100 + i)
        ref.append((F.copy(), dF.copy()))
    got = list(obj.stream(batches, seeds=[100 + i for i in range(len(batches))]))
    assert len(got) == len(ref)
    for (F, dF), (Fr, dFr) in zip(got, ref):
        assert np.array_equal(F, Fr) and np.array_equal(dF, dFr)
    # against the oracle too (entropy on the dumped device stream), first batch, first restart
    if Ns > 0:
        eps = va.default_engine().ctx.rng_dump(D, K, 1, Ns, 100)[0]
        gpo = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=p["meanfun"])
        o = R.negelcbo_vbmc(batches[0][:, 0], 0, vp, gpo, Ns, True, 0, eps=eps)
        assert abs(got[0][0][0] - o["F"]) < 1e-9 * max(1.0, abs(o["F"]))
        assert np.max(np.abs(got[0][1][:, 0] - o["dF"])) < 1e-8 * max(1.0, np.max(np.abs(o["dF"])))


def test_other_calls_between_submit_and_collect(va):
    """A synchronous evaluation, a prediction and a posterior update issued while two passes are in flight queue behind them on
    the stream; the collected results are those of the submitted batches."""
    p, gp, vp, batches = setup(va, 5, 6, 80, 12, 3, 8)
    T = batches[0].shape[0]
    obj = va.PreparedObjective(T, 8, 0, vp, gp, 400, 0, None)
    ref = [tuple(x.copy() for x in obj(th, seed=7 + i)) for i, th in enumerate(batches[:2])]
    obj.submit(batches[0], seed=7, slot=0)
    obj.submit(batches[1], seed=8, slot=1)
    other = va.negelcbo_batch(batches[2], 0, vp, gp, 400, True, 0, seed=9)          # synchronous call on the same context
    Xs = np.random.default_rng(0).standard_normal((33, 6))
    pred = va.gplite_pred(gp, Xs, None, None, False)
    gp2 = va.gplite_post(p["hyp"], p["X"], p["y"] + 0.1, 1, p["meanfun"])
    F1, dF1 = obj.collect(1)                                                        # any order
    F0, dF0 = obj.collect(0)
    assert np.array_equal(F0, ref[0][0]) and np.array_equal(dF0, ref[0][1])
    assert np.array_equal(F1, ref[1][0]) and np.array_equal(dF1, ref[1][1])
    assert np.array_equal(other["F"], va.negelcbo_batch(batches[2], 0, vp, gp, 400, True, 0, seed=9)["F"])
    assert np.all(np.isfinite(pred[2])) and gp2["post"][0]["alpha"].shape == (80,)


def test_pipeline_misuse_is_refused(va):
    p, gp, vp, batches = setup(va, 6, 3, 25, 4, 2, 4)
    T = batches[0].shape[0]
    obj = va.PreparedObjective(T, 4, 0, vp, gp, 100, 0, None)
    ctx = va.default_engine().ctx
    with pytest.raises(va.VbmcHipError, match="nothing submitted"):
        obj.collect(0)
    obj.submit(batches[0], seed=1, slot=0)
    with pytest.raises(va.VbmcHipError, match="uncollected"):
        obj.submit(batches[1], seed=2, slot=0)
    with pytest.raises(va.VbmcHipError, match="slot must be 0 .. 3"):
        obj.submit(batches[1], seed=2, slot=4)
    F, dF = obj.collect(0)
    F = F.copy()
    assert np.all(np.isfinite(F))
    # round 4: four slots on two streams (slot s on stream s & 1, two passes deep): all four in flight at once, collected in any order,
    # give the bits of the blocking call
    refs = [va.negelcbo_batch(batches[i & 1], 0, vp, gp, 100, True, 0, seed=50 + i, outputs=("F", "dF")) for i in range(4)]
    for i in range(4):
        obj.submit(batches[i & 1], seed=50 + i, slot=i)
    for i in (2, 0, 3, 1):
        Fi, dFi = obj.collect(i)
        assert np.array_equal(Fi, refs[i]["F"]) and np.array_equal(dFi, refs[i]["dF"])
    # per-component outputs are not offered through the pipelined form
    a = obj._slot(0)[0]
    b = type(a).from_buffer_copy(a)
    b.separate_K = 1
    b.compute_grad = 0
    with pytest.raises(va.VbmcUnsupported):
        ctx.check(ctx.lib.vbmc_elbo_submit(ctx.h, obj.dgp.h, C.byref(b), 0))
    # a failed submit leaves the slot free
    bad = batches[0].copy()
    bad[0, 0] = np.nan
    with pytest.raises(va.VbmcHipError, match="non-finite"):
        obj.submit(bad, seed=3, slot=0)
    obj.submit(batches[0], seed=1, slot=0)
    F2, _ = obj.collect(0)
    assert np.array_equal(F2, F)


def test_collect_with_another_layout_is_refused(va):
    """collecting with args whose optimize flags (a smaller T) or compute_grad differ from the submitted ones would
    copy plan.T doubles per restart into the caller's shorter arrays.  Refused; the slot stays collectable with the right args."""
    p, gp, vp, batches = setup(va, 7, 3, 25, 4, 2, 3)
    T = batches[0].shape[0]
    obj = va.PreparedObjective(T, 3, 0, vp, gp, 60, 0, None)
    ctx = va.default_engine().ctx
    obj.submit(batches[0], seed=4, slot=1)
    a = obj._slot(1)[0]
    b = type(a).from_buffer_copy(a)
    b.optimize[3] = 0                       # without the eta block: T - K
    with pytest.raises(va.VbmcHipError, match="differ from the submitted"):
        ctx.check(ctx.lib.vbmc_elbo_collect(ctx.h, C.byref(b), 1))
    b = type(a).from_buffer_copy(a)
    b.compute_grad = 0
    with pytest.raises(va.VbmcHipError, match="differ from the submitted"):
        ctx.check(ctx.lib.vbmc_elbo_collect(ctx.h, C.byref(b), 1))
    F, dF = obj.collect(1)
    ref = va.negelcbo_batch(batches[0], 0, vp, gp, 60, True, 0, seed=4)
    assert np.array_equal(F, ref["F"]) and np.array_equal(dF, ref["dF"])


def test_log_joint_role_of_the_entropy_launch_matches_the_separate_kernel():
    """Small grids (S R < half the compute units, value + gradient, no per-hyper-sample outputs): the expected log joint runs as
    extra single-wave workgroups of the MFMA entropy kernel's launch (entropy_mfma.h CO = true).
    The role is on in this process (VBMC_LJ_CO unset); asking for the per-component records (separate_K) or for the variance
    routes the SAME evaluation through the separate k_logjoint launch: G and its gradient agree to summation order, and the
    role's answer is pinned by the oracle like every other."""
    import vbmc_amd as va
    from oracle import vbmc_ref as R
    from tests._cases import synth_problem

    for (D, N, K, S, Ns, nR, seed) in [(10, 400, 50, 20, 56, 1, 3), (4, 70, 17, 3, 40, 2, 4), (14, 130, 36, 5, 24, 1, 5), (2, 30, 2, 1, 100, 3, 6),
                                       (7, 64, 21, 6, 30, 1, 7)]:
        p = synth_problem(seed, D, N, K, S)
        gp = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=p["meanfun"])
        vp = R.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
        vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
        theta = np.concatenate([p["mu"].reshape(-1, order="F"), np.log(p["sigma"]), np.log(p["lam"]), p["eta"]])
        th = np.asfortranarray(theta[:, None] + 0.03 * np.random.default_rng(seed).standard_normal((theta.size, nR)))
        co = va.negelcbo_batch(th, 0, vp, gp, Ns, True, 0, seed=11, outputs=("F", "G", "dG", "H", "dH"))
        sep = va.negelcbo_batch(th, 0, vp, gp, Ns, True, 2, seed=11, outputs=("F", "G", "dG", "H", "dH", "varG"))   # variance: separate launch
        assert np.array_equal(co["H"], sep["H"]) and np.array_equal(co["dH"], sep["dH"])       # same entropy kernel body, same stream
        assert np.allclose(co["G"], sep["G"], rtol=1e-13, atol=0)
        assert np.max(np.abs(co["dG"] - sep["dG"])) <= 1e-12 * np.max(np.abs(sep["dG"]))
        for r in range(nR):
            o = R.negelcbo_vbmc(th[:, r], 0, vp, gp, 0, True, 0)
            assert abs(co["G"][r] - o["G"]) <= 1e-10 * abs(o["G"])
            assert np.max(np.abs(co["dG"][:, r] - np.asarray(o["dG"]).reshape(-1))) <= 1e-9 * np.max(np.abs(o["dG"]))


def test_slot_streams_are_placed_by_measurement_in_a_process_with_other_queues(va):
    """abi_elbo.hip: stream_beside -- a fresh context in a process that already holds other hardware queues (those of other contexts
    here; tools/archive/r4_place_check.py does it with torch streams) creates its slot streams by measuring which candidates dispatch beside each
    other (tools/stream_pipes.hip); the results do not depend on it, and a second context goes through the same procedure."""
    keep = [va.Engine(0) for _ in range(3)]     # six more streams (a context's own and its second one each) ahead of the slot streams
    p, gp, vp, batches = setup(va, 11, 6, 60, 9, 3, 8)
    T = batches[0].shape[0]
    ref_obj = va.PreparedObjective(T, 8, 0, vp, gp, 400, 0, None)
    ref = [tuple(x.copy() for x in ref_obj(th, seed=40 + i)) for i, th in enumerate(batches)]
    for _ in range(2):
        eng = va.Engine(0)
        try:
            gp2 = va.gplite_post(p["hyp"], p["X"], p["y"], 1, p["meanfun"], engine=eng)
            obj = va.PreparedObjective(T, 8, 0, vp, gp2, 400, 0, None, engine=eng)
            got = list(obj.stream(batches, seeds=[40 + i for i in range(len(batches))]))
            for (F0, dF0), (F1, dF1) in zip(ref, got):
                assert np.array_equal(F0, F1) and np.array_equal(dF0, dF1)
        finally:
            eng.close() if hasattr(eng, "close") else None


def test_abandoned_generator_and_surrogate_freed_under_a_pass_in_flight(va):
    """(a) a stream() generator dropped half way leaves no slot busy (vbmc_elbo_abandon in its finally block) -- the next
    stream() over the same objective runs; (b) a pooled surrogate freed, and its blocks handed out again to a new upload, while a pass
    that reads it is still in flight on a slot stream: vbmc_gp_free waits for that stream first, so the pass completes on the OLD data
    and collects the bits of the blocking call."""
    p, gp, vp, batches = setup(va, 11, 10, 120, 50, 4, 16)
    T = batches[0].shape[0]
    obj = va.PreparedObjective(T, 16, 0, vp, gp, 2000, 0, None)
    ref = [tuple(x.copy() for x in obj(th, seed=30 + i)) for i, th in enumerate(batches)]
    g = obj.stream(batches, seeds=[30 + i for i in range(len(batches))])
    F0, dF0 = next(g)                       # four submitted, one collected: three passes in flight
    assert np.array_equal(F0, ref[0][0])
    g.close()                               # GeneratorExit -> finally -> abandon
    ctx = va.default_engine().ctx
    for sl in range(4):
        with pytest.raises(va.VbmcHipError, match="nothing submitted"):
            obj.collect(sl)
        obj.abandon(sl)                     # idle slot: a no-op
    got = list(obj.stream(batches, seeds=[30 + i for i in range(len(batches))]))
    for (F, dF), (Fr, dFr) in zip(got, ref):
        assert np.array_equal(F, Fr) and np.array_equal(dF, dFr)
    # (b)
    eng = va.default_engine()
    obj.submit(batches[1], seed=31, slot=0)
    obj.submit(batches[2], seed=32, slot=1)
    eng.invalidate()                                   # the engine's cache lets go of the surrogate ...
    obj.dgp.close()                                    # ... and its blocks go back to the pool while two passes read them
    gp2 = va.gplite_post(p["hyp"], p["X"] * 1.5 + 0.3, p["y"] - 2.0, 1, p["meanfun"])    # same sizes: the pool hands the same blocks out again
    _ = va.negelcbo_batch(batches[0], 0, vp, gp2, 2000, True, 0, seed=1)
    F1, dF1 = obj.collect(0)
    F2, dF2 = obj.collect(1)
    assert np.array_equal(F1, ref[1][0]) and np.array_equal(dF1, ref[1][1])
    assert np.array_equal(F2, ref[2][0]) and np.array_equal(dF2, ref[2][1])
