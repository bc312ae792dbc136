"""GPU parity: Bayesian-quadrature variance of the expected log joint (gplogjoint.m:273-413).

J_jk = nf_jk - z_k' K^-1 z_j cancels to ~1e-4..1e-8 of its terms, so J / varG are compared at
rel 1e-7 of the *term* scale (the oracle-vs-mpmath test uses 1e-8 for the same reason).
"""
import numpy as np
import pytest

from oracle import vbmc_ref as R
from tests._cases import golden_cases, load_golden, theta_from_inputs, vp_from_inputs
from tests.test_gpu_elbo import problem, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def va():
    import vbmc_amd

    return vbmc_amd


@pytest.mark.parametrize("path", golden_cases())
def test_golden_J_and_variance(va, path):
    inp, exp = load_golden(path)
    vp = vp_from_inputs(inp)
    gp = R.gplite_post(inp["hyp"], inp["X"], inp["y"], meanfun=inp["meanfun"])
    for s, post in enumerate(gp["post"]):
        post["alpha"] = np.array(exp["alpha"][s])
        post["L"] = np.array(exp["L"][s])
    theta = theta_from_inputs(inp)
    r = va.negelcbo_batch(theta, 0, vp, gp, 0, False, 1, separate_K=True)
    J = np.array(exp["J_sjk"])
    assert relerr(r["J_sjk"][:, :, :, 0], J) < 1e-8
    ref = R.negelcbo_vbmc(theta, 0, vp, gp, 0, False, 1, separate_K=True)
    assert relerr(r["varG"][0], ref["varG"]) < 1e-7
    assert relerr(r["varGss"][0], ref["varGss"]) < 1e-7
    r2 = va.negelcbo_batch(theta, 0, vp, gp, 0, False, 2, separate_K=True)
    ref2 = R.negelcbo_vbmc(theta, 0, vp, gp, 0, False, 2, separate_K=True)
    assert relerr(r2["varG"][0], ref2["varG"]) < 1e-7


@pytest.mark.parametrize("cfg", [(4, 60, 7, 3), (10, 150, 20, 4), (2, 33, 3, 1)])
def test_full_variance_eval_fullelcbo_form(va, cfg):
    """The eval_fullelcbo call: 11 outputs, full variance, separate_K (vpoptimize_vbmc.m:288-289)."""
    D, N, K, S = cfg
    p, gp, vp, theta = problem(11, D, N, K, S)
    Ns = 128
    eps = np.random.default_rng(2).standard_normal((K, Ns // 2, D))
    out = va.negelcbo_vbmc(theta, 0, vp, gp, Ns, 0, 1, nargout=11, eps=eps)
    F, dF, G, H, varF, dH, varGss, varG, varH, I_sk, J_sjk = out
    ref = R.negelcbo_vbmc(theta, 0, vp, gp, Ns, False, 1, separate_K=True, eps=eps)
    assert relerr(F, ref["F"]) < 1e-10 and relerr(G, ref["G"]) < 1e-10 and relerr(H, ref["H"]) < 1e-10
    assert relerr(I_sk, ref["I_sk"]) < 1e-10
    scale = max(1.0, np.max(np.abs(ref["J_sjk"])))
    assert np.max(np.abs(J_sjk - ref["J_sjk"])) < 1e-7 * scale
    assert abs(varG - ref["varG"]) < 1e-7 * max(1.0, abs(ref["varG"]), scale)
    assert abs(varGss - ref["varGss"]) < 1e-7 * max(1.0, abs(ref["varGss"]), scale)
    assert varH == 0.0 and varF == varG


def test_diag_variance_gradient_with_beta(va):
    """beta ~= 0 with gradient needs compute_var == 2 (negelcbo_vbmc.m:21-24,126-130)."""
    p, gp, vp, theta = problem(12, 5, 70, 6, 3)
    Ns = 64
    eps = np.random.default_rng(4).standard_normal((6, Ns // 2, 5))
    beta = 1.7
    F, dF, G, H, varF = va.negelcbo_vbmc(theta, beta, vp, gp, Ns, 1, 2, nargout=5, eps=eps)
    ref = R.negelcbo_vbmc(theta, beta, vp, gp, Ns, True, 2, eps=eps)
    assert relerr(varF, ref["varF"]) < 1e-7
    assert relerr(F, ref["F"]) < 1e-8
    assert relerr(dF, ref["dF"]) < 1e-6  # d sqrt(varF) amplifies the J cancellation
    # single hyper-sample: no between-sample term
    p, gp, vp, theta = problem(13, 3, 40, 4, 1)
    F, dF = va.negelcbo_vbmc(theta, beta, vp, gp, 0, 1, 2)
    ref = R.negelcbo_vbmc(theta, beta, vp, gp, 0, True, 2)
    assert relerr(F, ref["F"]) < 1e-8 and relerr(dF, ref["dF"]) < 1e-6


def test_low_noise_branch_Lchol_false(va):
    """gp.post(s).Lchol == false stores L = -inv(K + sn2 I) (gplite_core.m:98; gplogjoint.m:279,321)."""
    p, gp, vp, theta = problem(14, 4, 50, 5, 3)
    N = 50
    for s in (0, 2):  # mixed: samples 0 and 2 in the low-noise representation
        post = gp["post"][s]
        sl = 1.0 / post["sW"][0] ** 2
        Kinv = R.solve_upper(post["L"], R.solve_upper_t(post["L"], np.eye(N))) / sl
        post["L"] = -Kinv
        post["Lchol"] = False
    ref = R.negelcbo_vbmc(theta, 0, vp, gp, 0, False, 1, separate_K=True)
    out = va.negelcbo_vbmc(theta, 0, vp, gp, 0, 0, 1, nargout=11)
    scale = max(1.0, np.max(np.abs(ref["J_sjk"])))
    assert np.max(np.abs(out[10] - ref["J_sjk"])) < 1e-7 * scale
    assert abs(out[7] - ref["varG"]) < 1e-7 * max(1.0, scale)
    beta = 0.8
    F, dF = va.negelcbo_vbmc(theta, beta, vp, gp, 0, 1, 2)
    ref2 = R.negelcbo_vbmc(theta, beta, vp, gp, 0, True, 2)
    assert relerr(F, ref2["F"]) < 1e-8 and relerr(dF, ref2["dF"]) < 1e-6


def test_variance_paths_at_large_parameter_counts(va):
    """T = D K + 2 K + D in the thousands (found by the wide random sweep): the variance kernels size their LDS by T only
    when a gradient is wanted; the variance gradient beyond ~4000 parameters is refused cleanly, and a refused call leaves
    no stale device error behind for the next one."""
    p, gp, vp, theta = problem(7, 28, 150, 100, 2)          # T = 3028
    for cv, grad in ((1, False), (2, False), (2, True)):
        ref = R.negelcbo_vbmc(theta, 1.0 if cv == 2 else 0.0, vp, gp, 0, grad, cv)
        out = va.negelcbo_batch(theta, 1.0 if cv == 2 else 0.0, vp, gp, 0, grad, cv)
        assert relerr(out["varG"][0], ref["varG"]) < 1e-6 and relerr(out["F"][0], ref["F"]) < 1e-8
        if grad:
            assert relerr(out["dF"][:, 0], ref["dF"]) < 1e-6
    p2, gp2, vp2, theta2 = problem(7, 32, 60, 141, 1)       # T = 4826
    with pytest.raises(va.VbmcUnsupported):
        va.negelcbo_batch(theta2, 1.0, vp2, gp2, 10, True, 2, seed=1)
    out = va.negelcbo_batch(theta2, 0.0, vp2, gp2, 10, True, 0, seed=1)    # the next call is unaffected
    assert np.isfinite(out["F"][0]) and np.all(np.isfinite(out["dF"]))
