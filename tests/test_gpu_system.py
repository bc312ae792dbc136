"""GPU system test in the spirit of the reference's own pass criterion (test/runtest_vbmc.m:9,87: |ELBO - lnZ| < 0.5 and
posterior-mean RMSE < 0.5): every stage of the path in sequence, on the device, for a target whose answer is known.

  target     a normalised correlated-free Gaussian log density in D = 3 (lnZ = 0, known mean)
  GP fit     MAP hyper-parameters by L-BFGS on gplite_nlZ and its gradient (vbmc_gp_nlz)
  posterior  gplite_post, one rank-one append of a held-out point (vbmc_gp_rank1_update)
  VP fit     vpsieve_vbmc (batched entlb sieve) + vpoptimize_vbmc (on-device Adam, eval_fullelcbo with the BQ variance)
No oracle is involved: the checks are the analytic lnZ and mean, with the reference's own tolerances."""
import numpy as np
import pytest
import scipy.optimize

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def va():
    import vbmc_amd

    return vbmc_amd


def test_gaussian_target_elbo_and_mean(va):
    rng = np.random.default_rng(2024)
    D, N, K = 3, 90, 2
    mu_t = np.array([0.5, -0.3, 0.2])
    sig_t = np.array([1.0, 0.7, 1.3])
    Xall = mu_t + 1.4 * sig_t * rng.standard_normal((N + 1, D))
    yall = -0.5 * np.sum(((Xall - mu_t) / sig_t) ** 2, axis=1) - np.sum(np.log(sig_t)) - 0.5 * D * np.log(2 * np.pi)
    X, y = Xall[:N], yall[:N]
    gpd = {"X": X, "y": y, "s2": None, "covfun": 1, "Ncov": D + 1, "noisefun": (1, 0, 0), "Nnoise": 1, "meanfun": 4,
           "Nmean": 2 * D + 1, "meanfun_extras": None, "intmeanfun": 0}
    # ---- GP hyper-parameters: [log ell (D); log sf; log sn; m0; xm (D); log omega (D)]
    h0 = np.concatenate([np.zeros(D), [np.log(np.std(y))], [np.log(1e-2)], [np.max(y)], np.mean(X, axis=0), np.log(np.std(X, axis=0))])
    lo = np.concatenate([np.full(D, -3.0), [-6.0], [np.log(1e-3)], [-np.inf], np.full(D, -np.inf), np.full(D, -3.0)])
    hi = np.concatenate([np.full(D, 3.0), [6.0], [np.log(1.0)], [np.inf], np.full(D, np.inf), np.full(D, 3.0)])

    def obj(h):
        f, g = va.gplite_nlZ(h, gpd)
        return float(f), np.asarray(g, dtype=np.float64)

    res = scipy.optimize.minimize(obj, h0, jac=True, method="L-BFGS-B", bounds=list(zip(lo, hi)), options={"maxiter": 60})
    assert res.fun < obj(h0)[0] - 1.0                     # the device gradient is a descent direction that gets somewhere
    hyp = res.x.reshape(-1, 1)
    # ---- posterior, plus the held-out point by a rank-one append on the device
    gp = va.gplite_post(hyp, X, y, 1, 4)
    gp = va.gplite_post_rank1(gp, Xall[N], yall[N])
    assert gp["X"].shape == (N + 1, D)
    fmu = np.asarray(va.gplite_pred(gp, mu_t[None, :] + 0.3, None, None, False)[2]).reshape(-1)
    ytrue = -0.5 * np.sum((0.3 / sig_t) ** 2) - np.sum(np.log(sig_t)) - 0.5 * D * np.log(2 * np.pi)
    assert abs(fmu[0] - ytrue) < 0.05                      # the surrogate has learnt the log density
    # ---- variational posterior
    idx = rng.permutation(N)[:K]
    vp = va.make_vp(X[idx].T.copy(), np.full(K, 0.5), np.ones(D), eta=np.zeros(K))
    vp["w"] = np.full(K, 1.0 / K)
    vp2, varss, _ = va.vpoptimize_vbmc(30, 2, vp, gp, options={"MaxIterStochastic": 500}, rng=np.random.default_rng(1))
    st = vp2["stats"]
    post_mean = vp2["mu"] @ vp2["w"]
    rmse = float(np.sqrt(np.mean((post_mean - mu_t) ** 2)))
    assert abs(st["elbo"] - 0.0) < 0.5, st["elbo"]          # test/runtest_vbmc.m:87 (lnZ = 0 for a normalised density)
    assert rmse < 0.5, (post_mean, mu_t)                    # test/runtest_vbmc.m:9
    assert st["elbo_sd"] < 0.5 and np.isfinite(varss)


def test_profiling_modes_time_the_dominant_kernel_and_leave_the_results_alone(va):
    """vbmc_ctx_set_profiling (include/vbmc_hip.h): 1 = the call as it runs (a blocking call forks the log joint beside the kernel),
    2 = the kernel alone; both report a positive duration and neither changes a bit of the results."""
    import numpy as np

    from tests._cases import synth_problem

    p = synth_problem(5, 10, 200, 50, 8)
    eng = va.Engine(0)
    gp = va.gplite_post(p["hyp"], p["X"], p["y"], 1, p["meanfun"], engine=eng)
    vp = va.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    theta = np.concatenate([p["mu"].reshape(-1, order="F"), np.log(p["sigma"]), np.log(p["lam"]), p["eta"]])
    th = np.asfortranarray(theta[:, None] + 0.05 * np.random.default_rng(5).standard_normal((theta.size, 32)))
    outs = []
    for mode in (False, 1, 2):
        eng.ctx.set_profiling(mode)
        o = va.negelcbo_batch(th, 0, vp, gp, 2000, True, 0, seed=9, engine=eng, outputs=("F", "dF"))
        if mode:
            ent_ms, lj_ms = eng.ctx.last_kernel_ms()
            assert ent_ms > 0.0 and lj_ms > 0.0
        outs.append((o["F"].copy(), o["dF"].copy()))
    eng.ctx.set_profiling(False)
    for F, dF in outs[1:]:
        assert np.array_equal(F, outs[0][0]) and np.array_equal(dF, outs[0][1])
