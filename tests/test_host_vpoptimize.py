"""CPU: the host logic of vbmc_amd.optimize.vpoptimize_vbmc (chain selection, batching of the chains and of their
eval_fullelcbo calls, best-slot choice, the pruning loop with its I_sk / J_sjk bookkeeping) against the sequential oracle
restatement of misc/vpoptimize_vbmc.m.  The device entry point is replaced by an oracle-backed stand-in (this test runs
where there is no GPU); the same comparison runs on the real device in tests/test_gpu_vpoptimize.py."""
import numpy as np
import pytest

from oracle import vbmc_ref as R
from tests._vpopt import OPTS, compare, eps_from_trace, vpopt_problem


def _stream(D):
    def stream(seed, r, R_, K, Ns):
        return np.random.default_rng([int(seed) & 0xFFFFFFFF, int(seed) >> 32, r]).standard_normal((K, (Ns + 1) // 2, D))
    return stream


def _fake_batch(D, gp_any=None):
    """negelcbo_batch made of the oracle; gp = None (entropy-only evaluation: vpoptimize_vbmc's pruning attempts take the expected
    log joint from the cached per-component terms) evaluates with ``gp_any`` and hands back the entropy alone."""
    stream = _stream(D)

    def negelcbo_batch(thetas, beta, vp, gp, Ns=0, compute_grad=True, compute_var=None, thetabnd=None, *, separate_K=False,
                       seed=0, engine=None, outputs=None, **kw):
        thetas = np.asarray(thetas, dtype=np.float64)
        if thetas.ndim == 1:
            thetas = thetas[:, None]
        T, Rn = thetas.shape
        K = vp["K"]
        if gp is None:
            gp = gp_any
        out = {k: [] for k in ("F", "dF", "G", "H", "varG", "varGss", "I_sk", "J_sjk")}
        for r in range(Rn):
            eps = stream(seed, r, Rn, K, Ns) if Ns > 0 else None
            o = R.negelcbo_vbmc(thetas[:, r], beta, vp, gp, Ns, bool(compute_grad), int(compute_var or 0), thetabnd=thetabnd,
                                separate_K=separate_K, eps=eps)
            out["F"].append(o["F"]); out["G"].append(o["G"]); out["H"].append(o["H"])
            out["varG"].append(o["varG"]); out["varGss"].append(o["varGss"])
            if compute_grad:
                out["dF"].append(o["dF"])
            if separate_K:
                out["I_sk"].append(o["I_sk"])
                if compute_var:
                    out["J_sjk"].append(o["J_sjk"])
        res = {k: np.array(out[k]) for k in ("F", "G", "H", "varG", "varGss")}
        if compute_grad:
            res["dF"] = np.stack(out["dF"], axis=1)
        if separate_K:
            res["I_sk"] = np.stack(out["I_sk"], axis=2)
            if compute_var:
                res["J_sjk"] = np.stack(out["J_sjk"], axis=3)
        return res

    return negelcbo_batch


@pytest.mark.parametrize("nslow,midpoint", [(2, True), (1, True), (3, False)])
def test_vpoptimize_host_logic_matches_sequential_reference(monkeypatch, nslow, midpoint):
    import vbmc_amd.optimize as opt

    p, gp, vp = vpopt_problem()
    D = vp["D"]
    monkeypatch.setattr(opt, "negelcbo_batch", _fake_batch(D, gp))
    opts = dict(OPTS, ELCBOmidpoint=midpoint)
    trace = []
    vpa, varss_a, pruned_a = opt.vpoptimize_vbmc(12, nslow, vp, gp, options=opts, rng=np.random.default_rng(3), seed=5,
                                                 device_adam=False, trace=trace)
    vpb, varss_b, pruned_b = R.vpoptimize_vbmc(12, nslow, vp, gp, options=opts, rng=np.random.default_rng(3),
                                               eps_for=eps_from_trace(trace, _stream(D)))
    compare(vpa, vpb, varss_a, varss_b, pruned_a, pruned_b, 1e-12)
    assert sum(t["kind"] == "adam" for t in trace) == nslow
    assert sum(t["kind"] == "full" for t in trace) == nslow * (2 if midpoint else 1)
    nprune = sum(t["kind"] == "prune" for t in trace)
    assert nprune >= 1 and 0 < pruned_a <= nprune     # the loop ran and accepted at least one pruning
    if pruned_a:
        K0 = vp["K"]
        assert vpa["stats"]["J_sjk"].shape == (len(gp["post"]), K0, K0 - pruned_a)   # :239 deletes the third dimension only


def test_lockstep_adam_equals_independent_chains():
    """fminadam_lockstep: R chains sharing one batched objective call == each chain run alone by utils/fminadam.m's loop."""
    import vbmc_amd.optimize as opt

    rng = np.random.default_rng(0)
    A = rng.standard_normal((6, 6))
    A = A @ A.T + np.eye(6)
    noise = rng.standard_normal((400, 3))

    def f_single(x, r, it):
        return float(0.5 * x @ A @ x + 0.01 * noise[it - 1, r]), A @ x

    X0 = rng.standard_normal((6, 3)) * np.array([1.0, 0.01, 3.0])     # very different convergence times

    def fun_batch(X, it):
        vals = [f_single(X[:, r], r, it) for r in range(3)]
        return np.array([v[0] for v in vals]), np.stack([v[1] for v in vals], axis=1)

    xo, fo, xl, fl, its = opt.fminadam_lockstep(fun_batch, X0, 1e-3, 400)
    assert len(set(int(i) for i in its)) > 1
    for r in range(3):
        cnt = [0]

        def fun(x, r=r):
            cnt[0] += 1
            return f_single(x, r, cnt[0])

        x1, f1, xt1, ft1, it1 = opt.fminadam(fun, X0[:, r], None, None, 1e-3, 400)
        assert it1 == its[r]
        assert np.allclose(xt1, xl[r], rtol=0, atol=1e-14) and np.allclose(ft1, fl[r], rtol=0, atol=1e-14)
        assert np.allclose(x1, xo[:, r], rtol=0, atol=1e-14) and abs(f1 - fo[r]) < 1e-14


def test_cmaes_batched_minimises_a_noisy_quadratic():
    import vbmc_amd.optimize as opt

    rng = np.random.default_rng(1)
    for N in (8, 260):                      # full-covariance and separable variants
        c = rng.standard_normal(N)
        sc = np.exp(rng.uniform(-1, 1, N))
        nz = np.random.default_rng(2)

        def fun_batch(X):
            return np.sum((sc[:, None] * (X - c[:, None])) ** 2, axis=0) + 1e-6 * nz.standard_normal(X.shape[1])

        x, info = opt.cmaes_batched(fun_batch, np.zeros(N), np.ones(N), TolX=1e-6, TolFun=1e-4, TolHistFun=1e-5,
                                    MaxFunEvals=40000 if N < 100 else 150000, rng=np.random.default_rng(3))
        assert np.sqrt(np.mean((x - c) ** 2)) < (2e-2 if N < 100 else 0.25), (N, info["stop"], info["generations"])
