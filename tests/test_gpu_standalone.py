"""GPU parity of the path's pieces called on their own, as the reference allows: entmc_vbmc / entlb_vbmc
(ent/entmc_vbmc.m:1, ent/entlb_vbmc.m:1) without a surrogate and gplogjoint (misc/gplogjoint.m:1, called directly at
private/activesample_vbmc.m:155) without an entropy term, with every grad_flags subset the reference accepts.
fp64; tolerance 1e-10 on values, 1e-9 on gradients (summation order)."""
import numpy as np
import pytest

from oracle import vbmc_ref as R
from tests._cases import synth_problem

pytestmark = pytest.mark.gpu


from tests._cases import relerr  # noqa: E402  (relative to the largest reference entry; no floor)


@pytest.fixture(scope="module")
def va():
    import vbmc_amd

    return vbmc_amd


def make(seed, D, N, K, S):
    p = synth_problem(seed, D, N, K, S)
    gp = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=p["meanfun"], noisefun=p["noisefun"], s2=p["s2"])
    vp = R.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    return p, gp, vp


FLAGS = [(1, 1, 1, 1), (1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1), (1, 1, 0, 0), (0, 1, 1, 1), (1, 0, 1, 0)]


@pytest.mark.parametrize("flags", FLAGS)
def test_entmc_alone(va, flags):
    p, gp, vp = make(3, 5, 40, 7, 2)
    Ns = 90
    eps = p["rng"].standard_normal((vp["K"], Ns // 2, vp["D"]))
    H, dH = va.entmc_vbmc(vp, Ns, flags, True, eps=eps)
    Ho, dHo = R.entmc_vbmc(vp, Ns, flags, True, eps=eps)
    assert relerr(H, Ho) < 1e-10
    assert dH.shape == np.asarray(dHo).reshape(-1).shape
    assert relerr(dH, np.asarray(dHo).reshape(-1)) < 1e-9
    # value only: one output, no gradient work
    assert relerr(va.entmc_vbmc(vp, Ns, None, True, nargout=1, eps=eps), Ho) < 1e-10


@pytest.mark.parametrize("flags", FLAGS)
def test_entlb_alone(va, flags):
    p, gp, vp = make(4, 6, 30, 9, 1)
    H, dH = va.entlb_vbmc(vp, flags, True)
    Ho, dHo = R.entlb_vbmc(vp, flags, True)[:2]
    assert relerr(H, Ho) < 1e-10
    assert relerr(dH, np.asarray(dHo).reshape(-1)) < 1e-9


def test_entropy_defaults_and_refusals(va):
    p, gp, vp = make(5, 3, 20, 4, 1)
    # grad_flags omitted with two outputs -> all four groups (entmc_vbmc.m:8-10)
    H, dH = va.entlb_vbmc(vp)
    assert dH.size == 3 * 4 + 4 + 3 + 4
    # K = 1: exact entropy (entlb_vbmc.m:32-47); the Monte Carlo estimator has the same expectation
    vp1 = R.make_vp(p["mu"][:, :1], p["sigma"][:1], p["lam"], eta=np.zeros(1))
    vp1["w"] = np.ones(1)
    exact = 0.5 * 3 * (1 + np.log(2 * np.pi)) + 3 * np.log(vp1["sigma"][0]) + np.sum(np.log(vp1["lambda"]))
    assert relerr(va.entlb_vbmc(vp1, nargout=1), exact) < 1e-12
    assert abs(va.entmc_vbmc(vp1, 20000, nargout=1, seed=3) - exact) < 0.05
    # the ABI refuses variance outputs without a surrogate
    theta, _ = va.get_vptheta(vp)
    with pytest.raises(ValueError):
        va.negelcbo_batch(theta, 0.0, vp, None, 0, False, 1)


@pytest.mark.parametrize("flags", FLAGS)
def test_gplogjoint_alone(va, flags):
    p, gp, vp = make(6, 4, 35, 6, 3)
    F, dF = va.gplogjoint(vp, gp, flags, nargout=2)
    o = R.gplogjoint(vp, gp, flags, True, True, 0)
    assert relerr(F, o["F"]) < 1e-10
    assert relerr(dF, np.asarray(o["dF"]).reshape(-1)) < 1e-9


@pytest.mark.parametrize("cv", [1, 2])
def test_gplogjoint_variance_and_components(va, cv):
    p, gp, vp = make(7, 3, 25, 5, 2)
    o = R.gplogjoint(vp, gp, (0, 0, 0, 0), True, True, cv, separate_K=True)
    F, dF, varF, dvarF, varss, I_sk, J_sjk = va.gplogjoint(vp, gp, (0, 0, 0, 0), True, True, cv, nargout=7)
    assert dF.size == 0 and dvarF is None
    assert relerr(F, o["F"]) < 1e-10
    assert relerr(varF, o["varF"]) < 1e-8
    assert relerr(varss, o["varss"]) < 1e-8
    assert relerr(I_sk, o["I_sk"]) < 1e-10
    assert relerr(J_sjk, o["J_sjk"]) < 1e-8
    # three outputs: compute_var defaults to nargout > 2 (gplogjoint.m:14)
    F3, _, v3 = va.gplogjoint(vp, gp, (0, 0, 0, 0), nargout=3)
    assert relerr(v3, R.gplogjoint(vp, gp, (0, 0, 0, 0), True, True, 1)["varF"]) < 1e-8


def test_gplogjoint_refusals(va):
    p, gp, vp = make(8, 3, 20, 4, 2)
    # (round 5: the per-hyper-sample variance gradient and dvarF without the Jacobians are accelerated now:
    #  test_gplogjoint_variance_gradient_every_form)
    assert va.gplogjoint(vp, gp, True, False, True, 2, nargout=4)[3].shape[1] == 2
    assert va.gplogjoint(vp, gp, True, True, False, 2, nargout=4)[3].ndim == 1
    with pytest.raises(ValueError, match="FullVarianceGradient"):
        va.gplogjoint(vp, gp, True, True, True, 1, nargout=4)    # gplogjoint.m:27-30


@pytest.mark.parametrize("flags", FLAGS)
def test_untransformed_gradients(va, flags):
    """JACOBIAN_FLAG = 0 (round 3: on the device): gradients with respect to sigma, lambda and the weights w themselves --
    misc/gplogjoint.m:352-373, ent/entmc_vbmc.m:110-125 and ent/entlb_vbmc.m:132-143 skipped -- for every subset of the groups."""
    p, gp, vp = make(21, 4, 30, 6, 2)
    Ns = 60
    eps = p["rng"].standard_normal((vp["K"], Ns // 2, vp["D"]))
    H, dH = va.entmc_vbmc(vp, Ns, flags, False, eps=eps)
    Ho, dHo = R.entmc_vbmc(vp, Ns, flags, False, eps=eps)
    assert relerr(H, Ho) < 1e-10 and relerr(dH, np.asarray(dHo).reshape(-1)) < 1e-9
    H, dH = va.entlb_vbmc(vp, flags, False)
    Ho, dHo = R.entlb_vbmc(vp, flags, False)[:2]
    assert relerr(H, Ho) < 1e-10 and relerr(dH, np.asarray(dHo).reshape(-1)) < 1e-9
    F, dF = va.gplogjoint(vp, gp, flags, True, False, nargout=2)
    o = R.gplogjoint(vp, gp, flags, True, False, 0)
    assert relerr(F, o["F"]) < 1e-10 and relerr(dF, np.asarray(o["dF"]).reshape(-1)) < 1e-9
    # and they differ from the transformed ones wherever a Jacobian acts (every group but mu)
    if any(flags[1:]):
        _, dFj = va.gplogjoint(vp, gp, flags, True, True, nargout=2)
        assert np.max(np.abs(dFj - dF)) > 1e-6


@pytest.mark.parametrize("flags", [(1, 1, 1, 1), (1, 0, 1, 0), (0, 1, 0, 1)])
def test_gplogjoint_variance_gradient_output(va, flags):
    """[F,dF,varF,dvarF] = gplogjoint(vp,gp,grad_flags,1,1,2): the gradient of the diagonal variance as the 4th output
    (misc/gplogjoint.m:27,375-413; round 3: vbmc_elbo_args.dvarG)."""
    p, gp, vp = make(22, 3, 28, 5, 3)
    F, dF, varF, dvarF = va.gplogjoint(vp, gp, flags, True, True, 2, nargout=4)
    o = R.gplogjoint(vp, gp, flags, True, True, 2, compute_vargrad=True)
    assert relerr(F, o["F"]) < 1e-10 and relerr(dF, np.asarray(o["dF"]).reshape(-1)) < 1e-9
    assert relerr(varF, o["varF"]) < 1e-8
    ref = np.asarray(o["dvarF"]).reshape(-1)
    assert dvarF.shape == ref.shape
    assert np.max(np.abs(dvarF - ref)) < 1e-7 * max(1e-30, np.max(np.abs(ref)))


@pytest.mark.parametrize("cfg", [(4, 50, 6, 5), (3, 30, 3, 1), (10, 120, 20, 8)])
def test_gplogjoint_per_hyper_sample_outputs(va, cfg):
    """avg_flag = 0 (misc/gplogjoint.m:398-413 skipped): the call forms the reference's own callers use --
    [~,~,varF] = gplogjoint(vp,gp,0,0,0,1) at private/activesample_vbmc.m:155 and
    [~,~,~,~,~,I_sk,J_sjk] = gplogjoint(vp,gp,0,0,0,1,1) at misc/vpoptimizeweights_vbmc.m:42."""
    D, N, K, S = cfg
    p, gp, vp = make(17, D, N, K, S)
    ref = R.gplogjoint(vp, gp, grad_flags=0, avg_flag=False, jacobian_flag=False, compute_var=1, separate_K=True)
    F, dF, varF = va.gplogjoint(vp, gp, 0, 0, 0, 1, nargout=3)
    assert np.shape(F) == np.shape(ref["F"]) and np.size(dF) == 0
    assert relerr(F, ref["F"]) < 1e-10
    assert np.max(np.abs(np.asarray(varF) - ref["varF"])) < 1e-7 * max(1.0, np.max(np.abs(ref["varF"])))
    out = va.gplogjoint(vp, gp, 0, 0, 0, 1, 1, nargout=7)
    assert out[4] == 0.0 or S == 1                      # varss is only formed when averaging (:398-404)
    assert relerr(out[5], ref["I_sk"]) < 1e-10
    assert np.max(np.abs(out[6] - ref["J_sjk"])) < 1e-7 * max(1.0, np.max(np.abs(ref["J_sjk"])))
    # diagonal approximation and value only
    refd = R.gplogjoint(vp, gp, grad_flags=0, avg_flag=False, compute_var=2)
    Fd, _, varFd = va.gplogjoint(vp, gp, 0, 0, 0, 2, nargout=3)
    assert relerr(Fd, refd["F"]) < 1e-10 and np.max(np.abs(np.asarray(varFd) - refd["varF"])) < 1e-7 * max(1.0, np.max(np.abs(refd["varF"])))
    F0 = va.gplogjoint(vp, gp, 0, 0, nargout=1)
    assert relerr(F0, ref["F"]) < 1e-10
    # averaging the per-sample outputs on the host reproduces the averaged call (:399-411)
    Fa, _, varFa, _, varss = va.gplogjoint(vp, gp, 0, 1, 1, 1, nargout=5)
    if S > 1:
        Fs, vs = np.asarray(F), np.asarray(varF)
        assert abs(np.mean(Fs) - Fa) < 1e-12 * max(1.0, abs(Fa))
        vss = np.var(Fs, ddof=1)
        assert abs(np.mean(vs) + vss - varFa) < 1e-9 * max(1.0, abs(varFa))
        assert abs(vss + np.std(vs, ddof=1) - varss) < 1e-9 * max(1.0, abs(varss))


@pytest.mark.parametrize("flags", FLAGS)
@pytest.mark.parametrize("jac", [True, False])
def test_gplogjoint_per_hyper_sample_gradients(va, flags, jac):
    """avg_flag = 0 WITH grad_flags (round 4, vbmc_elbo_args.dG_s): dF is T x S -- the per-sample gradients of misc/gplogjoint.m:206-271 with
    the Jacobians of :352-373 (or without: jacobian_flag = 0) and the averaging of :411 skipped -- for every grad_flags subset; their mean
    over the hyper-samples is the averaged call's dF."""
    p, gp, vp = make(19, 5, 45, 7, 4)
    ref = R.gplogjoint(vp, gp, flags, False, jac, 0)
    F, dF = va.gplogjoint(vp, gp, flags, False, jac, nargout=2)
    assert np.shape(F) == (4,) and relerr(F, ref["F"]) < 1e-10
    assert dF.shape == np.asarray(ref["dF"]).shape == (dF.shape[0], 4)
    assert relerr(dF, ref["dF"]) < 1e-9
    Fa, dFa = va.gplogjoint(vp, gp, flags, True, jac, nargout=2)
    assert relerr(np.mean(dF, axis=1), dFa) < 1e-12
    # with the variance: F, varF per sample beside the gradients -- with or without the Jacobians (round 5)
    F2, dF2, varF2 = va.gplogjoint(vp, gp, flags, False, jac, 2, nargout=3)
    ref2 = R.gplogjoint(vp, gp, flags, False, jac, 2)
    assert relerr(dF2, ref2["dF"]) < 1e-9 and np.max(np.abs(np.asarray(varF2) - ref2["varF"])) < 1e-7 * max(1.0, np.max(np.abs(ref2["varF"])))


@pytest.mark.parametrize("flags", FLAGS)
@pytest.mark.parametrize("jac", [True, False])
@pytest.mark.parametrize("avg", [True, False])
def test_gplogjoint_variance_gradient_every_form(va, flags, jac, avg):
    """Round 5: dvarF (compute_var = 2, nargout >= 4; misc/gplogjoint.m:286-304) in the two forms refused through round 4 -- without the
    Jacobians (jacobian_flag = 0: :375-396 skipped) and per hyper-sample (avg_flag = 0: T x S, the averaging of :407-409 skipped;
    vbmc_elbo_args.dvarG_s) -- and in the old one, for every grad_flags subset, against the oracle; the per-sample gradients average to
    the averaged call's through :407-409."""
    p, gp, vp = make(23, 5, 45, 7, 4)
    ref = R.gplogjoint(vp, gp, flags, avg, jac, 2, compute_vargrad=True)
    F, dF, varF, dvarF = va.gplogjoint(vp, gp, flags, avg, jac, 2, nargout=4)
    assert relerr(F, ref["F"]) < 1e-10 and relerr(dF, np.asarray(ref["dF"])) < 1e-9
    assert np.max(np.abs(np.asarray(varF) - ref["varF"])) < 1e-7 * max(1.0, np.max(np.abs(ref["varF"])))
    assert np.shape(dvarF) == np.shape(ref["dvarF"])
    assert relerr(dvarF, ref["dvarF"], scale=max(np.max(np.abs(ref["dvarF"])), 1e-6 * np.max(np.abs(ref["varF"])))) < 1e-6     # z' K^-1 z cancels against nf_kk
    if not avg:
        Fa, dFa, varFa, dvarFa = va.gplogjoint(vp, gp, flags, True, jac, 2, nargout=4)
        Fs, dFs = np.asarray(F), np.asarray(dF)
        S = Fs.size
        dvv = 2.0 * np.sum(Fs[None, :] * dFs, axis=1) / (S - 1) - 2.0 * np.mean(Fs) * np.sum(dFs, axis=1) / (S - 1)      # :408
        assert relerr(np.sum(dvarF, axis=1) / S + dvv, dvarFa, scale=max(np.max(np.abs(dvarFa)), 1e-12)) < 1e-9


@pytest.mark.parametrize("cv", [0, 1, 2])
def test_gplogjoint_components_together_with_gradients(va, cv):
    """separate_K WITH grad_flags (round 4): negelcbo_vbmc refuses the combination (misc/negelcbo_vbmc.m:57-59), gplogjoint itself does not
    (misc/gplogjoint.m:13): F, dF, varF, I_sk, J_sjk of one call against the oracle."""
    p, gp, vp = make(21, 4, 40, 6, 3)
    o = R.gplogjoint(vp, gp, (1, 1, 1, 1), True, True, cv, separate_K=True)
    if cv == 1:      # seven outputs with a gradient ask for dvarF: the full variance has none (misc/gplogjoint.m:27-30)
        with pytest.raises(ValueError, match="FullVarianceGradient"):
            va.gplogjoint(vp, gp, (1, 1, 1, 1), True, True, cv, True, nargout=7)
        return
    out = va.gplogjoint(vp, gp, (1, 1, 1, 1), True, True, cv, True, nargout=7)
    assert relerr(out[0], o["F"]) < 1e-10 and relerr(out[1], np.asarray(o["dF"]).reshape(-1)) < 1e-9
    assert relerr(out[5], o["I_sk"]) < 1e-10
    if cv:
        assert relerr(out[2], o["varF"]) < 1e-8
        assert np.max(np.abs(out[6] - o["J_sjk"])) < 1e-7 * max(1.0, np.max(np.abs(o["J_sjk"])))


def test_gplogjoint_per_hyper_sample_outputs_against_mpmath_vectors(va):
    """avg_flag = 0 pinned directly on the committed 50-digit vectors (tests/golden/mp_case*.json: G_s, varG_s_full,
    varG_s_diag, I_sk, J_sjk), not only on the NumPy oracle: the per-sample F and varF the device hands back for
    [F,~,varF] = gplogjoint(vp,gp,0,0,0,compute_var) (misc/gplogjoint.m:398 skipped)."""
    from tests._cases import golden_cases, load_golden, vp_from_inputs

    for path in golden_cases():
        inp, exp = load_golden(path)
        vp = vp_from_inputs(inp)
        gp = R.gplite_post(inp["hyp"], inp["X"], inp["y"], meanfun=inp["meanfun"])
        for s, post in enumerate(gp["post"]):
            post["alpha"] = np.array(exp["alpha"][s])
            post["L"] = np.array(exp["L"][s])
        out = va.gplogjoint(vp, gp, 0, 0, 0, 1, 1, nargout=7)
        F, varF, I_sk, J_sjk = np.atleast_1d(out[0]), np.atleast_1d(out[2]), out[5], out[6]
        assert relerr(F, exp["G_s"]) < 1e-11
        assert relerr(I_sk, np.array(exp["I_sk"])) < 1e-11
        assert relerr(J_sjk, np.array(exp["J_sjk"])) < 1e-8      # z' K^-1 z cancels against nf_jk
        assert relerr(varF, exp["varG_s_full"]) < 1e-8
        _, _, varFd = va.gplogjoint(vp, gp, 0, 0, 0, 2, nargout=3)
        assert relerr(np.atleast_1d(varFd), exp["varG_s_diag"]) < 1e-8
