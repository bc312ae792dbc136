"""The NumPy oracle against the committed 50-digit mpmath golden vectors (CPU only)."""
import numpy as np
import pytest

from oracle import vbmc_ref as R
from tests._cases import (acq_golden_cases, golden_cases, load_acq_golden, load_golden, load_nlz_golden, load_pen_golden, load_pred_golden,
                          nlz_golden_cases, pen_golden_cases, pred_golden_cases, synth_problem, vp_from_inputs)

RTOL = 1e-11  # fp64 restatement vs 50-digit evaluation; sums of <= ~100 terms


def close(a, b, rtol=RTOL, atol=1e-13):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(1.0, float(np.max(np.abs(b))))
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.max(np.abs(a - b)) if a.size else 0.0
    assert err <= rtol * scale + atol, (err, scale)


@pytest.mark.parametrize("path", golden_cases())
def test_entmc_matches_mpmath(path):
    inp, exp = load_golden(path)
    vp = vp_from_inputs(inp)
    H, dH = R.entmc_vbmc(vp, 2 * inp["Mh"], (1, 1, 1, 1), True, eps=inp["eps"])
    close(H, exp["entmc_H"])
    close(dH, exp["entmc_dH"])


@pytest.mark.parametrize("path", golden_cases())
def test_entlb_matches_mpmath(path):
    inp, exp = load_golden(path)
    vp = vp_from_inputs(inp)
    H, dH = R.entlb_vbmc(vp, (1, 1, 1, 1), True)
    close(H, exp["entlb_H"])
    close(dH, exp["entlb_dH"], rtol=1e-10)


@pytest.mark.parametrize("path", golden_cases())
def test_gp_post_pred_match_mpmath(path):
    inp, exp = load_golden(path)
    gp = R.gplite_post(inp["hyp"], inp["X"], inp["y"], meanfun=inp["meanfun"])
    for s, post in enumerate(gp["post"]):
        assert post["Lchol"] and post["sn2_mult"] == 1.0
        close(post["alpha"], exp["alpha"][s], rtol=1e-9)  # cond(K) amplifies rounding
        close(post["L"], exp["L"][s], rtol=1e-10)
    _, _, fmu, fs2 = R.gplite_pred(gp, inp["Xstar"], ssflag=True)
    close(fmu.T, exp["pred_fmu"], rtol=1e-9)
    close(fs2.T, exp["pred_fs2"], rtol=1e-9)


@pytest.mark.parametrize("path", golden_cases())
def test_gplogjoint_matches_mpmath(path):
    inp, exp = load_golden(path)
    vp = vp_from_inputs(inp)
    gp = R.gplite_post(inp["hyp"], inp["X"], inp["y"], meanfun=inp["meanfun"])
    # plug the 50-digit alpha/L so the comparison isolates gplogjoint itself
    for s, post in enumerate(gp["post"]):
        post["alpha"] = np.array(exp["alpha"][s])
        post["L"] = np.array(exp["L"][s])
    r = R.gplogjoint(vp, gp, (1, 1, 1, 1), avg_flag=False, jacobian_flag=True, compute_var=1, separate_K=True)
    S = inp["S"]
    close(np.atleast_1d(r["F"]), exp["G_s"])
    close(np.asarray(r["dF"]).reshape(-1, S, order="F").T if S > 1 else np.asarray(r["dF"])[None, :], exp["dG_s"])
    close(r["I_sk"], exp["I_sk"])
    close(r["J_sjk"], exp["J_sjk"], rtol=1e-8)  # z' K^-1 z cancels against nf_jk
    close(np.atleast_1d(r["varF"]), exp["varG_s_full"], rtol=1e-8)
    r2 = R.gplogjoint(vp, gp, (0, 0, 0, 0), avg_flag=False, compute_var=2, separate_K=True)
    close(np.atleast_1d(r2["varF"]), exp["varG_s_diag"], rtol=1e-8)


def test_averaging_over_hyper_samples():
    """misc/gplogjoint.m:399-413 against a direct evaluation from per-sample values."""
    inp, exp = load_golden(golden_cases()[1])
    vp = vp_from_inputs(inp)
    gp = R.gplite_post(inp["hyp"], inp["X"], inp["y"], meanfun=inp["meanfun"])
    per = R.gplogjoint(vp, gp, (1, 1, 1, 1), avg_flag=False, compute_var=2, compute_vargrad=True)
    avg = R.gplogjoint(vp, gp, (1, 1, 1, 1), avg_flag=True, compute_var=2, compute_vargrad=True)
    S = inp["S"]
    F, dF, vF, dvF = per["F"], per["dF"], per["varF"], per["dvarF"]
    Fbar = F.sum() / S
    varFss = ((F - Fbar) ** 2).sum() / (S - 1)
    close(avg["F"], Fbar)
    close(avg["dF"], dF.sum(axis=1) / S)
    close(avg["varF"], vF.sum() / S + varFss)
    close(avg["varss"], varFss + np.std(vF, ddof=1))
    dvv = 2 * (F[None, :] * dF).sum(axis=1) / (S - 1) - 2 * Fbar * dF.sum(axis=1) / (S - 1)
    close(avg["dvarF"], dvF.sum(axis=1) / S + dvv)


@pytest.mark.parametrize("path", nlz_golden_cases())
def test_nlz_matches_mpmath(path):
    """gplite_nlZ value against the 50-digit definition; the analytic gradient (Q-matrix formulas,
    gplite_core.m:236-275) against 50-digit central differences of that value."""
    gp, hyp, exp = load_nlz_golden(path)
    for s in range(hyp.shape[1]):
        nlZ, dnlZ = R.gplite_nlZ(hyp[:, s], gp)
        close(nlZ, exp["nlZ"][s])
        close(dnlZ, exp["dnlZ"][s], rtol=1e-10)


def test_hypprior_closed_forms():
    """gplite_hypprior.m: Gaussian / Student-t / flat entries against scipy.stats log-pdfs and their slopes."""
    from scipy import stats
    hyp = np.array([0.3, -1.2, 2.0, 0.7])
    hp = {"mu": np.array([0.0, -1.0, np.nan, 1.0]), "sigma": np.array([2.0, 0.5, 1.0, np.inf]), "df": np.array([np.inf, 3.0, 3.0, 0.0])}
    lp, dlp = R.gplite_hypprior(hyp, hp)
    ref = stats.norm.logpdf(0.3, 0.0, 2.0) + stats.t.logpdf((-1.2 + 1.0) / 0.5, 3.0) - np.log(0.5)
    close(lp, ref)
    e = 1e-6
    for i in range(4):
        d = np.zeros(4)
        d[i] = e
        fd = (R.gplite_hypprior(hyp + d, hp)[0] - R.gplite_hypprior(hyp - d, hp)[0]) / (2 * e)
        assert abs(fd - dlp[i]) < 1e-8
    assert dlp[2] == 0.0 and dlp[3] == 0.0


@pytest.mark.parametrize("path", acq_golden_cases())
def test_oracle_acquisition_vs_mpmath(path):
    """acqwrapper_vbmc restatement (acqf, acqflog, acqus, acqfsn2, acqviqr with Ctmp / fs2a from the oracle's own
    precomputation) against the 50-digit evaluation of the definitions (tests/golden/mp_acq_case*.json)."""
    vp, gp, Xs, st, exp = load_acq_golden(path)
    Xa = st["ActiveImportanceSampling"]["Xa"]
    S, Na = len(gp["post"]), Xa.shape[0]
    _, Ct = R.acq_is_precompute(gp, Xa)
    fs2a = np.asarray(R.gplite_pred(gp, Xa, None, None, True)[3]).reshape(Na, -1)
    st = dict(st, ActiveImportanceSampling={"Xa": Xa, "Ctmp_mat": Ct, "fs2a": fs2a, "lnw": np.zeros((S, Na))})
    for name in ("acqf", "acqflog", "acqus", "acqfsn2", "acqviqr"):
        acq, fbar, vtot = R.acqwrapper_vbmc(Xs, vp, gp, st, name)
        tol = 1e-9 if name == "acqviqr" else 1e-11
        assert np.max(np.abs(acq - exp[name]) / np.maximum(1e-300, np.abs(exp[name]))) < tol, name
        assert np.max(np.abs(fbar - exp["fbar"])) < 1e-11 and np.max(np.abs(vtot - exp["vtot"]) / exp["vtot"]) < 1e-10


def check_pred_against_golden(gp, out, exp, lchol_expected):
    """shared by the oracle test here and the device test (tests/test_gpu_gplite.py): alpha, ymu, ys2, fmu, fs2, lp of
    gplite_post + gplite_pred(gp,Xstar,ystar,s2star,1) against tests/golden/mp_pred_case*.json.  Relative to max(1, |.|);
    the low-noise case (min sn2 = 2.6e-7, condition number ~1e7) is compared at 1e-8, the others at 1e-11."""
    tol = 1e-11 if lchol_expected else 1e-8
    for s, post in enumerate(gp["post"]):
        assert bool(post["Lchol"]) == lchol_expected and post["sn2_mult"] == 1.0
        close(post["alpha"], exp["alpha"][s], rtol=tol)
    ymu, ys2, fmu, fs2, lp = out
    for got, key in ((ymu, "fmu"), (ys2, "ys2"), (fmu, "fmu"), (fs2, "fs2"), (lp, "lp")):
        close(np.asarray(got).reshape(exp[key].T.shape), exp[key].T, rtol=tol)   # S = 1: a column either way


@pytest.mark.parametrize("path", pred_golden_cases())
def test_pred_general_noise_matches_mpmath(path):
    """gplite_noisefun.m:176-210 (constant, provided / scaled s2, output-dependent term), both branches of
    gplite_core.m:67-100 and gplite_pred.m:60-127 with ystar, s2star and lp."""
    inp, exp = load_pred_golden(path)
    gp = R.gplite_post(inp["hyp"], inp["X"], inp["y"], meanfun=inp["meanfun"], noisefun=inp["noisefun"], s2=inp["s2"])
    out = R.gplite_pred(gp, inp["Xstar"], inp["ystar"], inp["s2star"], True, nargout=5)
    check_pred_against_golden(gp, out, exp, bool(np.min(exp["min_sn2"]) >= 1e-6))


def penalty_problem(vp):
    """a small surrogate for the penalty tests: the penalties are the difference of two negelcbo calls (with / without thetabnd)"""
    p = synth_problem(9, vp["D"], 12, vp["K"], 2)
    return R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=p["meanfun"])


@pytest.mark.parametrize("path", pen_golden_cases())
def test_penalties_match_mpmath(path):
    """vpbndloss.m + softbndloss.m directly, and the soft-bound + weight penalties as negelcbo_vbmc.m:136-164 adds them
    (difference of the calls with and without thetabnd; deterministic entropy), against tests/golden/mp_pen_case*.json."""
    vp, theta, tb, exp = load_pen_golden(path)
    L, dL = R.vpbndloss(theta, vp, tb, tb["TolCon"])
    close(L, exp["L_bnd"], rtol=1e-13)
    close(dL, exp["dL_bnd"], rtol=1e-13)
    gp = penalty_problem(vp)
    r1 = R.negelcbo_vbmc(theta, 0, vp, gp, 0, True, 0, thetabnd=tb)
    r0 = R.negelcbo_vbmc(theta, 0, vp, gp, 0, True, 0)
    sc = max(1.0, abs(r1["F"]))
    assert abs((r1["F"] - r0["F"]) - (exp["L_bnd"] + exp["L_w"])) < 1e-12 * sc
    assert np.max(np.abs((r1["dF"] - r0["dF"]) - (exp["dL_bnd"] + exp["dL_w"]))) < 1e-12 * max(1.0, np.max(np.abs(r1["dF"])))
