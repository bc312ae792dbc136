"""GPU: the callers of the objective -- batched sieve, Adam loop, vpoptimize -- against the oracle."""
import numpy as np
import pytest

from oracle import vbmc_ref as R
from tests.test_gpu_elbo import problem, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def va():
    import vbmc_amd

    return vbmc_amd


def test_sieve_order_is_index_identical(va):
    """vpsieve_vbmc.m:74-83: R sequential negelcbo calls + stable sort == one batched pass + the same sort."""
    p, gp, vp, _ = problem(31, 5, 80, 6, 4)
    rng = np.random.default_rng(7)
    vp0_vec, vp0_type, beta, cvar, NSentK, NSentKFast, fill = va.vpsieve_vbmc(30, 3, vp, gp, rng=rng)
    assert NSentKFast == 0 and not cvar and len(vp0_vec) == 30
    # re-evaluate every (already sorted) candidate sequentially with the oracle
    opts = dict(TolLength=1e-6, TolWeight=1e-2, TolConLoss=0.01, WeightPenalty=0.1)
    _, tb = R.vpbounds(vp, gp, opts)
    ref = []
    for v in vp0_vec:
        th, v2 = R.get_vptheta(v)
        ref.append(R.negelcbo_vbmc(th, 0, v2, gp, 0, False, 0, thetabnd=tb)["F"])
    ref = np.array(ref)
    assert relerr(np.sort(fill), ref) < 1e-10
    assert list(R.sieve_order(ref)) == list(range(30))  # oracle agrees the order is ascending: index-identical
    # MC-entropy sieve (NSentFast > 0) with the device stream is deterministic for a fixed seed
    a = va.vpsieve_vbmc(12, 3, vp, gp, options={"NSentFast": lambda K: 50 * K}, rng=np.random.default_rng(1), seed=3)
    b = va.vpsieve_vbmc(12, 3, vp, gp, options={"NSentFast": lambda K: 50 * K}, rng=np.random.default_rng(1), seed=3)
    assert np.array_equal(a[6], b[6])


def test_sieve_with_fixed_groups_that_differ_between_candidates(va):
    """Non-optimised groups are not part of theta; candidates that differ in one (here: fixed weights)
    must still each be evaluated with their own values (sub-batching in sieve_evaluate)."""
    from vbmc_amd.optimize import sieve_evaluate
    p, gp, vp, _ = problem(35, 4, 60, 5, 3)
    vp = dict(vp, optimize_weights=False)
    vp.pop("eta", None)
    rng = np.random.default_rng(2)
    cands = []
    for i in range(7):
        v = dict(vp, mu=vp["mu"] + 0.1 * rng.standard_normal(vp["mu"].shape))
        w = rng.dirichlet(np.ones(5)) if i % 3 else vp["w"]      # three distinct weight vectors + repeats
        cands.append(dict(v, w=w))
    vals, _ = sieve_evaluate(cands, gp, 0, False, 0.0, None)
    ref = []
    for v in cands:
        th, v2 = R.get_vptheta(v)
        ref.append(R.negelcbo_vbmc(th, 0, v2, gp, 0, False, 0)["F"])
    assert relerr(vals, np.array(ref)) < 1e-10


def test_adam_trajectory_matches_oracle(va):
    """utils/fminadam.m driven by the HIP objective vs by the oracle objective, same eps per iteration."""
    p, gp, vp, theta = problem(32, 4, 50, 5, 3)
    K, D, Ns = 5, 4, 40
    opts = dict(TolLength=1e-6, TolWeight=1e-2, TolConLoss=0.01, WeightPenalty=0.1)
    vpb, tb = R.vpbounds(vp, gp, opts)
    eps_all = np.random.default_rng(11).standard_normal((60, K, Ns // 2, D))
    cnt = {"a": 0, "b": 0}

    def fun_hip(x):
        e = eps_all[cnt["a"]]
        cnt["a"] += 1
        r = va.negelcbo_batch(x, 0, vpb, gp, Ns, True, 0, tb, eps=e)
        return float(r["F"][0]), r["dF"][:, 0]

    def fun_ref(x):
        e = eps_all[cnt["b"]]
        cnt["b"] += 1
        r = R.negelcbo_vbmc(x, 0, vpb, gp, Ns, True, 0, thetabnd=tb, eps=e)
        return r["F"], r["dF"]

    xa, fa, xta, fta, ita = va.fminadam(fun_hip, theta, None, None, 1e-3, 60)
    xb, fb, xtb, ftb, itb = R.fminadam(fun_ref, theta, TolFun=1e-3, MaxIter=60)
    assert ita == itb
    assert relerr(fta, ftb) < 1e-7 and relerr(xta, xtb) < 1e-7 and relerr(xa, xb) < 1e-7


def test_vpoptimize_improves_elbo(va):
    p, gp, vp, theta = problem(33, 3, 60, 3, 2, target="student")
    vp["optimize_weights"] = True
    f0 = va.negelcbo_vbmc(*(va.get_vptheta(vp)[0],), 0, va.get_vptheta(vp)[1], gp, 0, 0, 0, nargout=1)[0]
    vp2, varss, pruned = va.vpoptimize_vbmc(20, 2, vp, gp, options={"MaxIterStochastic": 200}, rng=np.random.default_rng(0))
    assert np.isfinite(vp2["stats"]["elbo"]) and vp2["stats"]["elbo_sd"] >= 0
    assert abs(np.sum(vp2["w"]) - 1) < 1e-12 and abs(np.sum(vp2["lambda"] ** 2) - 3) < 1e-9
    assert vp2["stats"]["elbo"] > -f0 - 1.0  # optimisation did not make things (much) worse than the start
    assert vp2["stats"]["I_sk"].shape == (2, 3) and vp2["stats"]["J_sjk"].shape == (2, 3, 3)


def test_device_adam_wide_batch_on_the_walking_launch(va, monkeypatch):
    """64 chains in lock-step at a sample count where each iteration's entropy pass is several rounds of waves: the on-device loop's passes
    are blocking-style evaluations and take the walking launch (entropy_mfma.h: WALK).  Against the same loop on the chunk grid
    (VBMC_ENT_WALK=0) the objective values agree to the order of summation over a component's partial records -- and are not the same bits,
    i.e. the walk did run."""
    p, gp, vp, theta = problem(35, 10, 60, 50, 2)
    opts = dict(TolLength=1e-6, TolWeight=1e-2, TolConLoss=0.01, WeightPenalty=0.1)
    vpb, tb = R.vpbounds(vp, gp, opts)
    x0 = theta[:, None] + 0.03 * np.random.default_rng(4).standard_normal((theta.size, 64))
    monkeypatch.delenv("VBMC_ENT_WALK", raising=False)
    xw, fw, xtw, ftw, itw = va.fminadam_device(x0, 0, vpb, gp, 2000, tb, 1e-3, 6, seed=9)
    monkeypatch.setenv("VBMC_ENT_WALK", "0")
    xc, fc, xtc, ftc, itc = va.fminadam_device(x0, 0, vpb, gp, 2000, tb, 1e-3, 6, seed=9)
    assert np.array_equal(itw, itc)
    assert relerr(xw, xc) < 1e-9 and relerr(np.asarray(fw), np.asarray(fc)) < 1e-9
    assert not np.array_equal(np.asarray(xw), np.asarray(xc))


def test_device_adam_equals_host_adam(va):
    """vbmc_adam_batch (whole loop on the device) vs utils/fminadam.m's loop on the host calling the same
    device objective with the same per-iteration seeds: identical stopping iteration, iterates to round-off."""
    p, gp, vp, theta = problem(34, 4, 50, 5, 3)
    opts = dict(TolLength=1e-6, TolWeight=1e-2, TolConLoss=0.01, WeightPenalty=0.1)
    vpb, tb = R.vpbounds(vp, gp, opts)
    Ns, seed, MaxIter = 60, 77, 200
    it = {"n": 0}

    def fun(x):
        it["n"] += 1
        r = va.negelcbo_batch(x, 0, vpb, gp, Ns, True, 0, tb, seed=seed + it["n"])
        return float(r["F"][0]), r["dF"][:, 0]

    xh, fh, xth, fth, ith = va.fminadam(fun, theta, None, None, 1e-3, MaxIter)
    xd, fd, xtd, ftd, itd = va.fminadam_device(theta, 0, vpb, gp, Ns, tb, 1e-3, MaxIter, seed=seed)
    assert int(itd[0]) == ith
    assert relerr(ftd[0], fth) < 1e-9 and relerr(xtd[0], xth) < 1e-9
    assert relerr(xd[:, 0], xh) < 1e-9 and abs(fd[0] - fh) < 1e-9 * max(1, abs(fh))
    # two chains in lock-step == the same chains run alone
    x2 = np.stack([theta, theta + 0.05], axis=1)
    xb, fb, xtb, ftb, itb = va.fminadam_device(x2, 0, vpb, gp, Ns, tb, 1e-3, 80, seed=seed)
    assert relerr(ftb[0][: min(80, ith)], fth[: min(80, ith)]) < 1e-9
