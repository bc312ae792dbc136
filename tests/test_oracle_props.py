"""Closed forms, finite differences and linear-algebra identities that pin the oracle (CPU)."""
import math

import numpy as np
import pytest

from oracle import vbmc_ref as R
from tests._cases import synth_problem


def build(seed=0, D=3, N=20, K=4, S=3, **kw):
    p = synth_problem(seed, D, N, K, S, **kw)
    gp = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=p["meanfun"], noisefun=p["noisefun"], s2=p["s2"])
    vp = R.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    theta = np.concatenate([p["mu"].reshape(-1, order="F"), np.log(p["sigma"]), np.log(p["lam"]), p["eta"]])
    return p, gp, vp, theta


def fd(f, x, h=1e-6):
    return np.array([(f(x + h * e) - f(x - h * e)) / (2 * h) for e in np.eye(x.size)])


def test_entropy_single_component_closed_form():
    # ent/entlb_vbmc.m:34 exact; entmc converges to it (2.4775 vs 2.4726 at M=1e5)
    D = 3
    vp = R.make_vp(np.zeros((D, 1)), [0.7], [1.2, 0.8, 1.0], eta=[0.0])
    vp["w"] = np.ones(1)
    H_exact = 0.5 * D * (1 + math.log(2 * math.pi)) + D * math.log(0.7) + math.log(1.2 * 0.8)
    H, _ = R.entlb_vbmc(vp)
    assert abs(H - H_exact) < 1e-14
    Hm, _ = R.entmc_vbmc(vp, 40000, rng=np.random.default_rng(1))
    assert abs(Hm - H_exact) < 0.03


def test_entlb_identical_components():
    # two identical components: gamma_nk equal -> H = -log(nconst (2 s^2)^(-D/2))
    D = 2
    mu = np.zeros((D, 2))
    vp = R.make_vp(mu, [0.5, 0.5], [1.0, 1.0], eta=[0.0, 0.0])
    vp["w"] = np.array([0.5, 0.5])
    H, _ = R.entlb_vbmc(vp)
    expect = -math.log((2 * math.pi) ** (-D / 2) * (2 * 0.25) ** (-D / 2))
    assert abs(H - expect) < 1e-14


def test_gplogjoint_alpha_zero_closed_form():
    # alpha = 0  =>  F = sum_k w_k (m0 + nu_k)   (misc/gplogjoint.m:169-174,203)
    p, gp, vp, _ = build(S=1)
    gp["post"][0]["alpha"][:] = 0.0
    r = R.gplogjoint(vp, gp)
    hyp = gp["post"][0]["hyp"]
    D = p["D"]
    m0 = hyp[D + 2]
    xm = hyp[D + 3 : D + 3 + D]
    om = np.exp(hyp[D + 3 + D :])
    nu = np.array([
        -0.5 * np.sum((vp["mu"][:, k] ** 2 + vp["sigma"][k] ** 2 * vp["lambda"] ** 2 - 2 * vp["mu"][:, k] * xm + xm**2) / om**2)
        for k in range(p["K"])
    ])
    assert abs(r["F"] - np.sum(vp["w"] * (m0 + nu))) < 1e-12


@pytest.mark.parametrize("meanfun", [0, 1, 4])
def test_gplogjoint_gradient_is_exact_derivative(meanfun):
    p, gp, vp, theta = build(seed=3, meanfun=meanfun)
    r = R.negelcbo_vbmc(theta, 0, vp, gp, 0, True, 0)
    g = fd(lambda t: R.negelcbo_vbmc(t, 0, vp, gp, 0, False, 0)["G"], theta)
    assert np.max(np.abs(g - r["dG"])) < 2e-8 * max(1, np.max(np.abs(g)))
    h = fd(lambda t: R.negelcbo_vbmc(t, 0, vp, gp, 0, False, 0)["H"], theta)
    assert np.max(np.abs(h - r["dH"])) < 2e-8 * max(1, np.max(np.abs(h)))


def test_diag_variance_gradient_is_exact_derivative():
    p, gp, vp, theta = build(seed=4)
    r = R.negelcbo_vbmc(theta, 1.5, vp, gp, 0, True, 2)
    g = fd(lambda t: R.negelcbo_vbmc(t, 1.5, vp, gp, 0, False, 2)["F"], theta)
    assert np.max(np.abs(g - r["dF"])) < 5e-8 * max(1, np.max(np.abs(g)))


def test_entmc_eta_block_is_exact_derivative_other_blocks_are_not():
    # only the weight block differentiates the MC value itself
    p, gp, vp, theta = build(seed=5)
    K = p["K"]
    eps = np.random.default_rng(9).standard_normal((K, 100, p["D"]))
    r = R.negelcbo_vbmc(theta, 0, vp, gp, 200, True, 0, eps=eps)
    h = fd(lambda t: R.negelcbo_vbmc(t, 0, vp, gp, 200, False, 0, eps=eps)["H"], theta)
    assert np.max(np.abs(h[-K:] - r["dH"][-K:])) < 2e-8
    assert np.max(np.abs(h[:-K] - r["dH"][:-K])) > 1e-4


def test_penalties_gradient():
    p, gp, vp, theta = build(seed=6)
    opts = dict(TolLength=1e-6, TolWeight=1e-2, TolConLoss=0.01, WeightPenalty=0.1)
    vpb, tb = R.vpbounds(vp, gp, opts)
    th = theta.copy()
    th[0] += 10.0  # push one mean outside its box
    th[-1] = -9.0  # and one weight below the bound
    r = R.negelcbo_vbmc(th, 0, vpb, gp, 0, True, 0, thetabnd=tb)
    r0 = R.negelcbo_vbmc(th, 0, vpb, gp, 0, True, 0)
    assert r["F"] > r0["F"]
    g = fd(lambda t: R.negelcbo_vbmc(t, 0, vpb, gp, 0, False, 0, thetabnd=tb)["F"], th, h=1e-7)
    assert np.max(np.abs(g - r["dF"])) < 1e-5 * max(1, np.max(np.abs(g)))


@pytest.mark.parametrize("noisy", [False, True])
def test_gp_post_identities(noisy):
    p, gp, vp, _ = build(seed=7, N=25, noisy=noisy)
    D, X, y = p["D"], p["X"], p["y"]
    for post in gp["post"]:
        hyp = post["hyp"]
        ell = np.exp(hyp[:D])
        sf2 = math.exp(2 * hyp[D])
        Kmat = sf2 * np.exp(-0.5 * R.sq_dist(X.T / ell[:, None]))
        sn2 = R.gplite_noisefun(hyp[D + 1 : D + 2], X, gp["noisefun"], y, gp["s2"])
        sn2v = np.broadcast_to(sn2, y.shape) * post["sn2_mult"]
        sl = float(np.min(sn2)) * post["sn2_mult"]
        assert post["Lchol"]
        L = post["L"]
        assert np.allclose(np.tril(L, -1), 0)
        assert np.max(np.abs(L.T @ L * sl - (Kmat + np.diag(sn2v)))) < 1e-9 * sf2  # L'L*sl = K + sn2 I
        m = R.gplite_meanfun(hyp[D + 2 :], X, gp["meanfun"])
        assert np.max(np.abs((Kmat + np.diag(sn2v)) @ post["alpha"] - (y - m))) < 1e-7
    _, _, fmu, fs2 = R.gplite_pred(gp, X, s2star=gp["s2"], ssflag=True)
    assert np.all(fs2 >= 0)
    if not noisy:
        assert np.max(np.abs(fmu - y[:, None])) < 0.05


def test_low_noise_branch_matches_cholesky_branch():
    # Lchol=false stores L = -inv(K+sn2 I); all consumers must agree with the Lchol path
    p, gp, vp, theta = build(seed=8, S=2)
    gp2 = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=4)
    for post in gp2["post"]:
        N = p["N"]
        sl = 1.0 / post["sW"][0] ** 2
        Kinv = R.solve_upper(post["L"], R.solve_upper_t(post["L"], np.eye(N))) / sl
        post["L"] = -Kinv
        post["Lchol"] = False
    a = R.negelcbo_vbmc(theta, 0, vp, gp, 0, False, 1, separate_K=True)
    b = R.negelcbo_vbmc(theta, 0, vp, gp2, 0, False, 1, separate_K=True)
    assert abs(a["varG"] - b["varG"]) < 1e-9 * abs(a["varG"])
    pa = R.gplite_pred(gp, p["X"][:5] + 0.1)
    pb = R.gplite_pred(gp2, p["X"][:5] + 0.1)
    assert np.max(np.abs(pa[3] - pb[3])) < 1e-8


def test_rank1_equals_full_posterior():
    # property pattern of gplite/gplite_test.m:87-105
    p, gp, _, _ = build(seed=9, N=18)
    gp_small = R.gplite_post(p["hyp"], p["X"][:-1], p["y"][:-1], meanfun=4)
    gp_r1 = R.gplite_post_rank1(gp_small, p["X"][-1], p["y"][-1])
    for a, b in zip(gp["post"], gp_r1["post"]):
        assert np.max(np.abs(a["alpha"] - b["alpha"])) < 1e-8 * np.max(np.abs(a["alpha"]))
        assert np.max(np.abs(a["L"] - b["L"])) < 1e-9 * np.max(np.abs(a["L"]))


def test_cholesky_jitter_retry():
    # duplicate training points + tiny noise force p>0 and the x10 retry (gplite_core.m:77-80)
    rng = np.random.default_rng(0)
    X = rng.standard_normal((6, 2))
    X = np.vstack([X, X])
    y = rng.standard_normal(12)
    hyp = np.array([0.0, 0.0, 3.0, math.log(1.0000001e-3), 0.0, 0.0, 0.0, 0.0, 0.0])[:, None]
    hyp[3, 0] = math.log(1.1e-3)  # sn2 = 1.21e-6 >= 1e-6 -> Lchol; K/sn2 ~ 4e8 -> rank-deficient
    gp = R.gplite_post(hyp, X, y, meanfun=4)
    assert gp["post"][0]["sn2_mult"] >= 1.0
    assert np.all(np.isfinite(gp["post"][0]["alpha"]))


def test_sq_dist_against_direct():
    rng = np.random.default_rng(1)
    a = rng.standard_normal((3, 7)) + 100.0
    b = rng.standard_normal((3, 5)) + 100.0
    C = R.sq_dist(a, b)
    direct = np.sum((a[:, :, None] - b[:, None, :]) ** 2, axis=0)
    assert np.max(np.abs(C - direct)) < 1e-10
    assert np.max(np.abs(R.sq_dist(a) - np.sum((a[:, :, None] - a[:, None, :]) ** 2, axis=0))) < 1e-10


def test_theta_roundtrip_and_rescale():
    p, gp, vp, theta = build(seed=2)
    th, vp2 = R.get_vptheta(vp)
    assert abs(np.sum(vp2["lambda"] ** 2) - p["D"]) < 1e-12  # rescale_params.m:28-30
    vp3 = R.rescale_params(vp2, th)
    assert np.allclose(vp3["mu"], vp2["mu"]) and np.allclose(vp3["sigma"], vp2["sigma"])
    assert np.allclose(vp3["w"], vp2["w"])


def test_sieve_order_is_matlab_stable_sort():
    v = np.array([3.0, 1.0, np.nan, 1.0, -2.0, np.inf])
    assert list(R.sieve_order(v)) == [4, 1, 3, 0, 5, 2]


def test_fminadam_quadratic():
    A = np.diag([1.0, 4.0, 9.0])

    def fun(x):
        return 0.5 * x @ A @ x, A @ x

    x, f, xtab, ftab, it = R.fminadam(fun, np.array([1.0, -1.0, 0.5]), MaxIter=2000)
    assert np.max(np.abs(x)) < 0.05 and it >= 40 and it % 20 == 0


def test_vbmc_pdf_and_acq_restatement():
    """vbmc_pdf (transformed space) against scipy's multivariate normal; acquisition formulas on a GP where the
    prediction is known: far from the data fs2 -> sf2 and fmu -> the mean function."""
    from scipy import stats

    rng = np.random.default_rng(0)
    D, K = 3, 4
    vp = {"K": K, "D": D, "mu": rng.standard_normal((D, K)), "sigma": 0.5 + rng.random(K), "lambda": np.array([0.7, 1.0, 1.4]),
          "w": np.array([0.1, 0.2, 0.3, 0.4])}
    X = rng.standard_normal((6, D))
    ref = sum(vp["w"][k] * stats.multivariate_normal.pdf(X, vp["mu"][:, k], np.diag((vp["sigma"][k] * vp["lambda"]) ** 2))
              for k in range(K))
    assert np.max(np.abs(R.vbmc_pdf_transformed(vp, X) - ref) / ref) < 1e-13
    Xt = rng.standard_normal((12, D))
    y = -0.5 * np.sum(Xt**2, axis=1)
    hyp = np.array([np.log(0.5)] * D + [np.log(1.3), np.log(1e-2), 0.4])[:, None]
    gp = R.gplite_post(hyp, Xt, y, meanfun=1)
    far = 50.0 + rng.standard_normal((4, D))
    st = {"ymax": float(np.max(y)), "VarianceRegularizedAcqFcn": True, "TolGPVar": 1e-4}
    acq, fbar, vtot = R.acqwrapper_vbmc(far, vp, gp, st, "acqflog")
    assert np.allclose(fbar, 0.4) and np.allclose(vtot, 1.3**2)
    p = np.maximum(R.vbmc_pdf_transformed(vp, far), R.REALMIN)
    assert np.allclose(acq, -(np.log(1.3**2) + 0.4 - st["ymax"] + np.log(p)))
    a2, _, _ = R.acqwrapper_vbmc(far, vp, gp, st, "acqf", outside=np.array([True, False, False, False]))
    assert np.isinf(a2[0]) and np.all(a2[1:] <= 0)


def test_iqr_lookahead_variance_identity():
    """The quantity inside acqviqr/acqimiqr, s_pred^2 = fs2a - C^2/ys2, is the GP posterior variance at the
    importance points after one more (noisy) observation at the candidate -- checked against an actual rank-one
    update of the oracle GP (constant noise, one hyper-sample), which shares no code with acq_iqr."""
    rng = np.random.default_rng(4)
    D, N, Na = 3, 25, 9
    X = rng.standard_normal((N, D))
    y = -0.5 * np.sum(X**2, axis=1)
    hyp = np.array([np.log(0.7)] * D + [np.log(1.1), np.log(0.15), 0.2])[:, None]
    gp = R.gplite_post(hyp, X, y, meanfun=1)
    sn2 = np.exp(2 * hyp[D + 1, 0])
    Xa = rng.standard_normal((Na, D))
    xs = rng.standard_normal((1, D))
    _, Ct = R.acq_is_precompute(gp, Xa)
    fs2a = np.asarray(R.gplite_pred(gp, Xa, None, None, True)[3]).reshape(Na, 1)
    gl = np.ones(D)
    gp2 = dict(gp, X_rescaled=X.copy(), sn2new=np.full(N, sn2))
    st = {"gplengthscale": gl, "ActiveImportanceSampling": {"Xa": Xa, "Ctmp_mat": Ct, "fs2a": fs2a, "lnw": np.zeros((1, Na))}}
    _, _, fmu, fs2 = R.gplite_pred(gp, xs, None, None, True)
    fmu = np.asarray(fmu).reshape(1, 1)
    fs2 = np.asarray(fs2).reshape(1, 1)
    val = R.acq_iqr("acqviqr", xs, None, gp2, st, fmu, fs2, fmu[:, 0], fs2[:, 0])
    gp_new = R.gplite_post_rank1(gp, xs, 0.123)
    s_new = np.sqrt(np.asarray(R.gplite_pred(gp_new, Xa, None, None, True)[3]).reshape(Na))
    u = R.ACQ_U
    zz = u * s_new + np.log1p(-np.exp(-2 * u * s_new))
    ref = np.log(np.sum(np.exp(zz - zz.max()))) + zz.max()
    assert abs(val[0] - ref) < 1e-9
