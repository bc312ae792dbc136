"""Regressions for the round-1 advisor findings, device vs oracle through the C ABI."""
import numpy as np
import pytest

from oracle import vbmc_ref as R
from tests._cases import synth_problem
from tests.test_gpu_elbo import relerr
from tests.test_gpu_nlz import make_gp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def va():
    import vbmc_amd

    return vbmc_amd


def _problem(seed=5, D=4, N=50, K=5, S=3):
    p = synth_problem(seed, D, N, K, S)
    gp = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=4)
    vp = R.make_vp(p["mu"], p["sigma"], 1.7 * p["lam"], eta=p["eta"])   # sum(lambda^2) != D: NOT normalised
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    return p, gp, vp


@pytest.mark.parametrize("flags", [(1, 0, 1, 0), (1, 1, 0, 0), (0, 0, 0, 1), (1, 1, 1, 1), (0, 1, 0, 0)])
def test_standalone_wrappers_with_unnormalised_lambda_and_partial_flags(va, flags):
    """ADVICE medium: _with_grad_groups built theta from the rescaled vp but passed the un-rescaled one for the fixed
    groups.  The reference functions read vp as it is (ent/entmc_vbmc.m:13-20, misc/gplogjoint.m:64-83): an un-normalised
    lambda is a legal input and the result must not depend on which groups carry a gradient."""
    p, gp, vp = _problem()
    Ns = 40
    eps = np.random.default_rng(2).standard_normal((vp["K"], Ns // 2, vp["D"]))
    H, dH = va.entmc_vbmc(vp, Ns, flags, True, eps=eps)
    Hr, dHr = R.entmc_vbmc(vp, Ns, grad_flags=flags, jacobian_flag=True, eps=eps)
    assert abs(H - Hr) < 1e-10 * max(1.0, abs(Hr)) and relerr(dH, dHr) < 1e-9
    H, dH = va.entlb_vbmc(vp, flags, True)
    Hr, dHr = R.entlb_vbmc(vp, grad_flags=flags, jacobian_flag=True)
    assert abs(H - Hr) < 1e-10 * max(1.0, abs(Hr)) and relerr(dH, dHr) < 1e-9
    G, dG = va.gplogjoint(vp, gp, flags, True, True, 0, nargout=2)
    ref = R.gplogjoint(vp, gp, grad_flags=flags, avg_flag=True, jacobian_flag=True, compute_var=0)
    assert abs(G - ref["F"]) < 1e-10 * max(1.0, abs(ref["F"])) and relerr(dG, ref["dF"]) < 1e-9
    # value only (nargout = 1) with some optimize_* flags off in the vp itself
    vq = dict(vp, optimize_lambda=False, optimize_sigma=False)
    assert abs(va.entlb_vbmc(vq, None, True, nargout=1) - R.entlb_vbmc(vq, grad_flags=False)[0]) < 1e-10
    assert abs(va.gplogjoint(vq, gp, None, nargout=1) - R.gplogjoint(vq, gp, grad_flags=False, compute_var=0)["F"]) < 1e-9


def test_nlz_unsupported_forms_raise_vbmc_unsupported(va):
    """ADVICE low: VbmcUnsupported was constructed with one argument (TypeError instead of the fall-through signal)."""
    gp, draw = make_gp(np.random.default_rng(0), 12, 2, 1, (1, 0, 0))
    with pytest.raises(va.VbmcUnsupported):
        va.gplite_nlZ(draw(), dict(gp, intmeanfun=1))


def test_rank1_s2_consistency(va):
    """ADVICE low: exactly one of gp.s2 / s2star set must be a clear error, not a TypeError from np.concatenate."""
    p = synth_problem(3, 3, 20, 2, 2)
    gp = va.gplite_post(p["hyp"], p["X"], p["y"], 1, 4, (1, 0, 0), None)
    with pytest.raises(ValueError):
        va.gplite_post_rank1(gp, np.zeros(3), 0.1, s2star=0.5)
    pn = synth_problem(3, 3, 20, 2, 2, noisy=True)
    gpn = va.gplite_post(pn["hyp"], pn["X"], pn["y"], 1, 4, pn["noisefun"], pn["s2"])
    with pytest.raises(ValueError):
        va.gplite_post_rank1(gpn, np.zeros(3), 0.1)
    g2 = va.gplite_post_rank1(gpn, np.zeros(3), 0.1, s2star=0.5)   # heteroskedastic: full update (gplite_post.m:76-79)
    assert g2["X"].shape[0] == 21 and g2["s2"].shape[0] == 21


@pytest.mark.parametrize("noisefun", [(1, 0, 1), (1, 1, 1), (1, 2, 1)])
def test_pred_and_acquisition_with_output_dependent_noise(va, noisefun):
    """ADVICE low: noisefun(3) = 1 was refused by every prediction / acquisition entry point although fmu / fs2 never
    depend on the test-point noise; ys2 honours ystar (gplite_noisefun.m:198-207) and skips the term when ystar = []."""
    rng = np.random.default_rng(11)
    D, N, S = 3, 40, 3
    gp0, draw = make_gp(rng, N, D, 4, noisefun)
    hyp = np.stack([draw() for _ in range(S)], axis=1)
    ref = R.gplite_post(hyp, gp0["X"], gp0["y"], meanfun=4, noisefun=noisefun, s2=gp0["s2"])
    gp = va.gplite_post(hyp, gp0["X"], gp0["y"], 1, 4, noisefun, gp0["s2"])
    Xs = 1.2 * rng.standard_normal((33, D))
    ys = np.median(gp0["y"]) + rng.standard_normal(33)      # some below the threshold, some above
    s2s = 0.02 + 0.03 * rng.random(33) if noisefun[1] else None
    for ystar in (None, ys):
        for ss in (True, False):
            o = va.gplite_pred(gp, Xs, ystar, s2s, ss)
            r = R.gplite_pred(ref, Xs, ystar, s2s, ssflag=ss)
            for x, z in zip(o, r):
                assert relerr(x, z) < 1e-8
    o5 = va.gplite_pred(gp, Xs, ys, s2s, False, nargout=5)
    r5 = R.gplite_pred(ref, Xs, ys, s2s, ssflag=False, nargout=5)
    assert relerr(o5[4], r5[4]) < 1e-8
    # the y-dependent term really is in ys2 (and only there)
    a = va.gplite_pred(gp, Xs, None, s2s, True)
    b = va.gplite_pred(gp, Xs, ys, s2s, True)
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.any(b[1] > a[1])
    # acquisition sweep on such a surrogate
    p = synth_problem(2, D, N, 3, S)
    vp = R.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    st = {"ymax": float(np.max(gp0["y"])), "VarianceRegularizedAcqFcn": True, "TolGPVar": 1e-4}
    acq = va.acqwrapper_vbmc(Xs, vp, gp, st, False, "acqf_vbmc", None)
    acr = R.acqwrapper_vbmc(Xs, vp, ref, st, "acqf")[0]
    assert relerr(acq, acr) < 1e-7


def test_entry_points_refuse_outputs_they_do_not_serve(va):
    """dvarG_s / dG_s (and the other per-hyper-sample / per-component outputs) are served by vbmc_elbo_batch alone.  The Adam
    loop and the sharded evaluation run elbo_plan, which validates the fields, and then neither allocate nor read them back: they used
    to return VBMC_OK with the caller's buffer untouched.  Now: VBMC_ERR_UNSUPPORTED, and the buffer is still untouched."""
    import ctypes as C

    from vbmc_amd._lib import VbmcUnsupported, ptr
    from vbmc_amd.elbo import _build_args

    p = synth_problem(4, 3, 30, 4, 2)
    gp = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=4)
    vp = R.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    theta = np.concatenate([p["mu"].reshape(-1, order="F"), np.log(p["sigma"]), np.log(p["lam"]), p["eta"]]).reshape(-1, 1)
    eng = va.default_engine()
    ctx = eng.ctx
    T, S = theta.shape[0], 2
    dgp = eng.device_gp(gp, need_L=True)
    for field in ("dvarG_s", "dG_s"):
        cv = 2 if field == "dvarG_s" else 0
        a, keep, _ = _build_args(np.asfortranarray(theta), 1.0 if cv else 0.0, vp, gp, 20, True, cv, None, False, None, None, False, 3, eng, 0.0)
        sentinel = np.full((T, S, 1), 12345.0, order="F")
        setattr(a, field, ptr(sentinel))
        x = np.zeros((T, 1), order="F"); f = np.zeros(1); it = np.zeros(1, dtype=np.int32)
        with pytest.raises(VbmcUnsupported):
            ctx.check(ctx.lib.vbmc_adam_batch(ctx.h, dgp.h, C.byref(a), 1e-3, 5, 1e-3, 0.1, 200.0, ptr(x), ptr(f),
                                               it.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None))
        assert np.all(sentinel == 12345.0)
        if cv == 0:
            n = C.c_size_t(0)
            with pytest.raises(VbmcUnsupported):
                ctx.check(ctx.lib.vbmc_elbo_shard_size(ctx.h, dgp.h, C.byref(a), 2, C.byref(n)))
