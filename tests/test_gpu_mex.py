"""GPU: the MEX gateway (matlab/vbmc_hip_mex.cpp) EXECUTED -- compiled against the functional mock of the mx* API
(tests/mock_mex/) and linked with libvbmc_hip.so -- command by command, MATLAB calling convention (nlhs / prhs order / struct
fields / output shapes), each output compared BIT FOR BIT with the same call made through the ctypes mirror (vbmc_amd): both
sit on the same C ABI and the library is deterministic, so any difference is a marshalling defect in the gateway.

The argument lists are the ones the .m shims under matlab/ pass (tests/test_matlab_static.py checks that the shims' calls
have these shapes)."""
import numpy as np
import pytest

from oracle import vbmc_ref as R
from tests._cases import synth_problem
from tests._mex import MexError

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mex():
    from tests import _mex

    m = _mex.mex()
    m.call(0, "open", 0)
    yield m
    assert m.live_arrays() == 0          # nothing the gateway created outlives its call (plhs handed back and freed)


@pytest.fixture(scope="module")
def va():
    import vbmc_amd

    return vbmc_amd


def make(seed=3, D=4, N=45, K=6, S=3, **kw):
    p = synth_problem(seed, D, N, K, S, **kw)
    gp = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=p["meanfun"], noisefun=p["noisefun"], s2=p["s2"])
    vp = R.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    theta = np.concatenate([p["mu"].reshape(-1, order="F"), np.log(p["sigma"]), np.log(p["lam"]), p["eta"]])
    return p, gp, vp, theta


def gp_struct(gp):
    """gp as MATLAB holds it (gplite_post.m:94-157): the fields the gateway reads."""
    return {"X": gp["X"], "y": gp["y"].reshape(-1, 1), "meanfun": gp["meanfun"], "noisefun": np.array(gp["noisefun"], dtype=float).reshape(1, -1),
            "Ncov": gp["Ncov"], "Nnoise": gp["Nnoise"],
            "post": [{"hyp": q["hyp"].reshape(-1, 1), "alpha": q["alpha"].reshape(-1, 1), "sW": np.asarray(q["sW"]).reshape(-1, 1),
                      "L": q["L"], "sn2_mult": q["sn2_mult"], "Lchol": bool(q["Lchol"])} for q in gp["post"]]}


def vp_struct(vp):
    """vp as MATLAB holds it (setupvars_vbmc.m:78-99): mu D x K, sigma / w / eta 1 x K, lambda D x 1."""
    return {"D": vp["D"], "K": vp["K"], "mu": vp["mu"], "sigma": vp["sigma"].reshape(1, -1), "lambda": vp["lambda"].reshape(-1, 1),
            "w": vp["w"].reshape(1, -1), "eta": vp["eta"].reshape(1, -1), "optimize_mu": bool(vp["optimize_mu"]),
            "optimize_sigma": bool(vp["optimize_sigma"]), "optimize_lambda": bool(vp["optimize_lambda"]),
            "optimize_weights": bool(vp["optimize_weights"]), "delta": np.zeros((1, 1))}


def bounds(vp, theta, rng):
    """thetabnd of misc/vpbounds.m: bounds on the EXTENDED vector [mu(:); ln sigma_k + ln lambda_d (D x K); eta], a few violated."""
    D, K = vp["D"], vp["K"]
    ext = np.concatenate([vp["mu"].reshape(-1, order="F"), (np.log(vp["sigma"])[None, :] + np.log(vp["lambda"])[:, None]).reshape(-1, order="F"),
                          vp["eta"]])
    n = ext.size
    assert n == 2 * D * K + K
    return {"lb": (ext - np.abs(rng.standard_normal(n)) * 0.3).reshape(-1, 1), "ub": (ext + np.abs(rng.standard_normal(n)) * 0.3 - 0.1).reshape(-1, 1),
            "TolCon": 0.01, "WeightThreshold": 0.15, "WeightPenalty": 0.3}


def same(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(a, b, equal_nan=True), float(np.nanmax(np.abs(a - b)))


def test_elbo_with_host_draws_all_twelve_outputs(mex, va):
    """'elbo' as matlab/negelcbo_vbmc.m and matlab/gplogjoint.m call it: parity-mode draws, bounds, variance, separate_K,
    per-hyper-sample outputs."""
    p, gp, vp, theta = make()
    S, K, D = 3, 6, 4
    rng = np.random.default_rng(0)
    tb = bounds(vp, theta, rng)
    Ns = 30
    eps = rng.standard_normal((K, Ns // 2, D))                    # ctypes layout (K, Ns/2, D) C-order
    eps_m = np.asfortranarray(eps.transpose(2, 1, 0))             # MATLAB's D x Ns/2 x K
    h = mex.call(1, "gp_upload", gp_struct(gp))[0]
    assert h.dtype == np.uint64 and h.shape == (1, 1)
    hh = np.uint64(h[0, 0])
    tbp = {k: (np.asarray(v).reshape(-1) if k in ("lb", "ub") else v) for k, v in tb.items()}
    # (a) value + gradient with the diagonal variance and its gradient (beta ~= 0 needs compute_var = 2), bounds, host draws
    out = mex.call(10, "elbo", hh, theta.reshape(-1, 1), vp_struct(vp), Ns, 1, 2, 0, 0.7, tb, eps_m, 0, S)
    F, dF, G, H, varG, dH, varGss, I_sk, J_sjk, dG = out
    ref = va.negelcbo_batch(theta, 0.7, vp, gp, Ns, True, 2, tbp, eps=eps, eps_shared=True)
    assert F.shape == (1, 1) and dF.shape == (theta.size, 1) and I_sk.size == 0 and J_sjk.size == 0
    same(F[0, 0], ref["F"][0]); same(G[0, 0], ref["G"][0]); same(H[0, 0], ref["H"][0])
    same(varG[0, 0], ref["varG"][0]); same(varGss[0, 0], ref["varGss"][0])
    same(dF[:, 0], ref["dF"][:, 0]); same(dH[:, 0], ref["dH"][:, 0]); same(dG[:, 0], ref["dG"][:, 0])
    # (b) value only with the full variance, the per-component terms (separate_K) and the per-hyper-sample outputs: 12 outputs
    out = mex.call(12, "elbo", hh, theta.reshape(-1, 1), vp_struct(vp), Ns, 0, 1, 1, 0, None, eps_m, 0, S)
    F, dF, G, H, varG, dH, varGss, I_sk, J_sjk, dG, G_s, varG_s = out
    ref = va.negelcbo_batch(theta, 0, vp, gp, Ns, False, 1, None, separate_K=True, eps=eps, eps_shared=True,
                            outputs=("F", "G", "H", "varG", "varGss", "I_sk", "J_sjk", "G_s", "varG_s"))
    assert dF.size == 0 and I_sk.shape == (S, K) and J_sjk.shape == (S, K, K) and G_s.shape == (1, S) and varG_s.shape == (1, S)
    same(F[0, 0], ref["F"][0]); same(G[0, 0], ref["G"][0]); same(H[0, 0], ref["H"][0]); same(varG[0, 0], ref["varG"][0])
    same(I_sk, ref["I_sk"][:, :, 0]); same(J_sjk, ref["J_sjk"][:, :, :, 0])
    same(G_s[0], ref["G_s"][:, 0]); same(varG_s[0], ref["varG_s"][:, 0])
    # value only, one output, no bounds, device RNG keyed by the seed argument
    (F1,) = mex.call(1, "elbo", hh, theta.reshape(-1, 1), vp_struct(vp), Ns, 0, 0, 0, 0, None, None, 1234, S)
    same(F1[0, 0], va.negelcbo_batch(theta, 0, vp, gp, Ns, False, 0, seed=1234)["F"][0])
    # entropy alone: NULL surrogate handle (matlab/entmc_vbmc.m)
    o = mex.call(6, "elbo", np.uint64(0), theta.reshape(-1, 1), vp_struct(vp), Ns, 1, 0, 0, 0, None, eps_m, 0, 1)
    re = va.negelcbo_batch(theta, 0, vp, None, Ns, True, 0, eps=eps, eps_shared=True)
    same(o[3][0, 0], re["H"][0]); same(o[5][:, 0], re["dH"][:, 0])
    # 14th argument: JACOBIAN_FLAG = 0 as the stand-alone shims pass it (matlab/entmc_vbmc.m, matlab/gplogjoint.m)
    o = mex.call(6, "elbo", np.uint64(0), theta.reshape(-1, 1), vp_struct(vp), Ns, 1, 0, 0, 0, None, eps_m, 0, 1, 1)
    rn = va.negelcbo_batch(theta, 0, vp, None, Ns, True, 0, eps=eps, eps_shared=True, jacobian_flag=False)
    same(o[3][0, 0], rn["H"][0]); same(o[5][:, 0], rn["dH"][:, 0])
    assert np.max(np.abs(rn["dH"][:, 0] - re["dH"][:, 0])) > 1e-8
    o = mex.call(10, "elbo", hh, theta.reshape(-1, 1), vp_struct(vp), 0, 1, 0, 0, 0, None, None, 0, S, 1)
    rn = va.negelcbo_batch(theta, 0, vp, gp, 0, True, 0, jacobian_flag=False)
    same(o[2][0, 0], rn["G"][0]); same(o[9][:, 0], rn["dG"][:, 0])
    # 13th output: the gradient of the diagonal variance (matlab/gplogjoint.m, dvarF)
    o = mex.call(13, "elbo", hh, theta.reshape(-1, 1), vp_struct(vp), 0, 1, 2, 0, 0, None, None, 0, S)
    rv = va.negelcbo_batch(theta, 0, vp, gp, 0, True, 2, outputs=("F", "dF", "G", "dG", "varG", "dvarG"))
    same(o[4][0, 0], rv["varG"][0]); same(o[12][:, 0], rv["dvarG"][:, 0])
    assert np.max(np.abs(o[12])) > 0
    # 14th output: the gradient per hyper-sample, T x S (gplogjoint with avg_flag = 0 and grad_flags; round 4)
    o = mex.call(14, "elbo", hh, theta.reshape(-1, 1), vp_struct(vp), 0, 1, 0, 0, 0, None, None, 0, S)
    rs = va.negelcbo_batch(theta, 0, vp, gp, 0, True, 0, outputs=("G", "dG", "G_s", "dG_s"))
    same(o[10][0, :], rs["G_s"][:, 0]); same(o[13], rs["dG_s"][:, :, 0])
    # 15th output (ABI 5): the variance gradient per hyper-sample, T x S, with and without the Jacobians (14th argument)
    for nojac in (0, 1):
        o = mex.call(15, "elbo", hh, theta.reshape(-1, 1), vp_struct(vp), 0, 1, 2, 0, 0, None, None, 0, S, nojac)
        rs = va.negelcbo_batch(theta, 0, vp, gp, 0, True, 2, outputs=("G", "dG", "varG", "dvarG", "dvarG_s", "dG_s"), jacobian_flag=not nojac)
        same(o[12][:, 0], rs["dvarG"][:, 0]); same(o[14], rs["dvarG_s"][:, :, 0]); same(o[13], rs["dG_s"][:, :, 0])
    assert o[13].shape == (theta.size, S) and np.max(np.abs(np.mean(o[13], axis=1) - rs["dG"][:, 0])) < 1e-12 * max(1.0, np.max(np.abs(rs["dG"])))
    with pytest.raises(MexError) as e:                            # the Jacobian of the soft bounds is part of the penalty's gradient
        mex.call(2, "elbo", hh, theta.reshape(-1, 1), vp_struct(vp), Ns, 1, 0, 0, 0, tb, eps_m, 0, S, 1)
    assert e.value.identifier in ("vbmc_hip:error", "vbmc_hip:unsupported")
    mex.call(0, "gp_free", hh)


def test_elbo_batch_sieve_and_fullelcbo_forms(mex, va):
    """'elbo_batch' as matlab/vbmc_hip_sieve.m (3 outputs) and matlab/vpoptimize_vbmc.m step 4 (8 outputs) call it."""
    p, gp, vp, theta = make(seed=4)
    S, K = 3, 6
    rng = np.random.default_rng(1)
    R_ = 5
    Th = np.asfortranarray(theta[:, None] + 0.05 * rng.standard_normal((theta.size, R_)))
    tb = bounds(vp, theta, rng)
    tbp = {k: (np.asarray(v).reshape(-1) if k in ("lb", "ub") else v) for k, v in tb.items()}
    hh = np.uint64(mex.call(1, "gp_upload", gp_struct(gp))[0][0, 0])
    F, dF, varG = mex.call(3, "elbo_batch", hh, Th, vp_struct(vp), 0, 0, 1, 0, tb, 77)
    ref = va.negelcbo_batch(Th, 0, vp, gp, 0, False, 1, tbp, seed=77)
    assert F.shape == (1, R_) and dF.size == 0
    same(F[0], ref["F"]); same(varG[0], ref["varG"])
    F, dF, varG, G, H, varGss, I_sk, J_sjk = mex.call(8, "elbo_batch", hh, Th, vp_struct(vp), 40, 0, 1, 0, None, 99, 1, S)
    ref = va.negelcbo_batch(Th, 0, vp, gp, 40, False, 1, None, separate_K=True, seed=99)
    assert I_sk.shape == (S, K, R_) and J_sjk.shape == (S, K, K, R_)
    same(F[0], ref["F"]); same(G[0], ref["G"]); same(H[0], ref["H"]); same(varG[0], ref["varG"]); same(varGss[0], ref["varGss"])
    same(I_sk, ref["I_sk"]); same(J_sjk, ref["J_sjk"])
    # with gradients (what a batched optimiser step reads)
    F, dF = mex.call(2, "elbo_batch", hh, Th, vp_struct(vp), 40, 1, 0, 0, tb, 5)
    ref = va.negelcbo_batch(Th, 0, vp, gp, 40, True, 0, tbp, seed=5)
    same(F[0], ref["F"]); same(dF, ref["dF"])
    mex.call(0, "gp_free", hh)


def test_adam_tables(mex, va):
    """'adam' as matlab/vpoptimize_vbmc.m step 3 calls it: x, f, iters, the iterate table T x MaxIter x R and ftab."""
    p, gp, vp, theta = make(seed=5)
    rng = np.random.default_rng(2)
    Th = np.asfortranarray(theta[:, None] + 0.02 * rng.standard_normal((theta.size, 2)))
    tb = bounds(vp, theta, rng)
    tbp = {k: (np.asarray(v).reshape(-1) if k in ("lb", "ub") else v) for k, v in tb.items()}
    hh = np.uint64(mex.call(1, "gp_upload", gp_struct(gp))[0][0, 0])
    MaxIter = 60
    x, f, it, xmid, xtab, ftab = mex.call(6, "adam", hh, Th, vp_struct(vp), 20, 0, 0, tb, 11, 0.001, MaxIter, np.array([0.001, 0.05, 200.0]))
    xr, fr, xt, ft, itr = va.fminadam_device(Th, 0, vp, gp, 20, tbp, TolFun=0.001, MaxIter=MaxIter,
                                             master_stepsize={"min": 0.001, "max": 0.05, "decay": 200.0}, seed=11)
    assert it.dtype == np.int32 and xtab.shape == (theta.size, MaxIter, 2) and ftab.shape == (MaxIter, 2)
    same(x, xr); same(f[0], fr); assert list(it[0]) == list(itr)
    for r in range(2):
        same(xtab[:, : itr[r], r], xt[r]); same(ftab[: itr[r], r], ft[r])
        same(xmid[:, r], xt[r][:, int(np.argmin(ft[r]))])            # the best midpoint, picked inside the library
    # the form matlab/vpoptimize_vbmc.m uses: four outputs, no tables
    x4, _, _, xmid4 = mex.call(4, "adam", hh, Th, vp_struct(vp), 20, 0, 0, tb, 11, 0.001, MaxIter, np.array([0.001, 0.05, 200.0]))
    same(x4, xr); same(xmid4, xmid)
    xq, fq, xmq, none, itq = va.fminadam_device(Th, 0, vp, gp, 20, tbp, TolFun=0.001, MaxIter=MaxIter,
                                                master_stepsize={"min": 0.001, "max": 0.05, "decay": 200.0}, seed=11, tables=False)
    assert none is None
    same(xq, xr); same(xmq, xmid)
    mex.call(0, "gp_free", hh)


@pytest.mark.parametrize("noisy", [False, True])
def test_gp_post_pred_rank1(mex, va, noisy):
    """'gp_post', 'gp_pred', 'gp_rank1' as matlab/gplite_post.m / gplite_pred.m call them."""
    p = synth_problem(6, 3, 40, 4, 3, noisy=noisy)
    nf = np.array(p["noisefun"], dtype=float).reshape(1, -1)
    s2 = None if p["s2"] is None else p["s2"].reshape(-1, 1)
    alpha, L, sW, mult, lch, h = mex.call(6, "gp_post", p["hyp"], p["X"], p["y"].reshape(-1, 1), s2, p["meanfun"], nf)
    gp = va.gplite_post(p["hyp"], p["X"], p["y"], 1, p["meanfun"], p["noisefun"], p["s2"])
    assert L.shape == (40, 40, 3) and lch.dtype == np.uint8 and h.dtype == np.uint64
    for s, q in enumerate(gp["post"]):
        same(alpha[:, s], q["alpha"]); same(L[:, :, s], q["L"]); same(sW[:, s], q["sW"])
        assert mult[s, 0] == q["sn2_mult"] and bool(lch[s, 0]) == q["Lchol"]
    hh = np.uint64(h[0, 0])
    Xs = np.asfortranarray(np.random.default_rng(3).standard_normal((37, 3)))
    s2s = np.full((37, 1), 0.5) if noisy else None
    for ss in (0, 1):
        outs = mex.call(4, "gp_pred", hh, Xs, None, s2s, ss, 3)
        ref = va.gplite_pred(gp, Xs, None, None if s2s is None else s2s.reshape(-1), bool(ss))
        for a, b in zip(outs, ref):
            same(a.reshape(b.shape, order="F"), b)
    if not noisy:
        xs = np.array([[0.3, -0.2, 0.1]])
        gp1 = va.gplite_post_rank1(gp, xs, 0.4)
        Xn = np.asfortranarray(np.vstack([p["X"], xs]))
        sn2_eff = np.array([1.0 / q["sW"][-1] ** 2 for q in gp1["post"]])
        # the shim passes sn2_eff = sn2 * sn2_mult per hyper-sample; recompute it the way gplite_post.m:207 does
        sn2_eff = np.array([np.exp(2.0 * q["hyp"][gp["Ncov"]]) * q["sn2_mult"] for q in gp["post"]])
        a1, L1, h1 = mex.call(3, "gp_rank1", hh, Xn, 0.4, None, None, sn2_eff.reshape(1, -1))
        for s, q in enumerate(gp1["post"]):
            same(a1[:, s], q["alpha"]); same(L1[:, :, s], q["L"])
        mex.call(0, "gp_free", np.uint64(h1[0, 0]))
    mex.call(0, "gp_free", hh)


def test_limits_command_and_a_posterior_that_stays_on_the_device(mex, va):
    """'limits' hands the shims the library's own numbers (matlab/vbmc_hip_supported.m); 'gp_post' asked for its sixth output leaves the
    posterior on the device and the objective evaluated on that handle equals the one on an uploaded copy WITHOUT any upload
    (matlab/gplite_post.m registers the handle with vbmc_hip_gp_handle: the next negelcbo_vbmc sends nothing; 'stats' counts)."""
    lim = mex.call(1, "limits")[0]
    assert isinstance(lim, dict) and list(lim) == ["max_D", "max_K", "max_N", "max_Na", "max_T_vargrad", "delta_ok", "meanfun"]
    assert (lim["max_D"][0, 0], lim["max_K"][0, 0], lim["max_N"][0, 0], lim["max_Na"][0, 0], lim["delta_ok"][0, 0]) == (32, 512, 9696, 256, 1)
    assert lim["meanfun"].reshape(-1).tolist() == [0, 1, 4]
    p, gp, vp, theta = make()
    S = 3
    n0 = mex.call(1, "stats")[0].reshape(-1)
    nf = np.array(p["noisefun"], dtype=float).reshape(1, -1)
    out = mex.call(6, "gp_post", p["hyp"], p["X"], p["y"].reshape(-1, 1), None, p["meanfun"], nf)
    hd = np.uint64(out[5][0, 0])
    n1 = mex.call(1, "stats")[0].reshape(-1)
    assert (n1 - n0).tolist() == [0, 1, 0]
    a = mex.call(2, "elbo", hd, theta.reshape(-1, 1), vp_struct(vp), 40, 1, 0, 0, 0, None, None, 77, S)
    n2 = mex.call(1, "stats")[0].reshape(-1)
    assert (n2 - n0).tolist() == [0, 1, 0]                  # the evaluation uploaded nothing
    hu = np.uint64(mex.call(1, "gp_upload", gp_struct(gp))[0][0, 0])
    b = mex.call(2, "elbo", hu, theta.reshape(-1, 1), vp_struct(vp), 40, 1, 0, 0, 0, None, None, 77, S)
    assert (mex.call(1, "stats")[0].reshape(-1) - n0).tolist() == [1, 1, 0]
    assert abs(a[0][0, 0] - b[0][0, 0]) <= 1e-12 * abs(b[0][0, 0]) and np.max(np.abs(a[1] - b[1])) <= 1e-11 * np.max(np.abs(b[1]))
    mex.call(0, "gp_free", hd)
    mex.call(0, "gp_free", hu)
    # five outputs: the gateway frees the device posterior itself (the round-5 shim's form: nothing leaks)
    mex.call(5, "gp_post", p["hyp"], p["X"], p["y"].reshape(-1, 1), None, p["meanfun"], nf)
    assert (mex.call(1, "stats")[0].reshape(-1) - n0).tolist() == [1, 1, 0]


def test_gp_nlz_and_sq_dist(mex, va):
    p = synth_problem(7, 3, 35, 4, 5)
    nf = np.array(p["noisefun"], dtype=float).reshape(1, -1)
    nlz, g = mex.call(2, "gp_nlz", p["hyp"], p["X"], p["y"].reshape(-1, 1), None, p["meanfun"], nf)
    gp0 = {"X": p["X"], "y": p["y"], "s2": None, "covfun": 1, "meanfun": p["meanfun"], "noisefun": p["noisefun"], "Ncov": 4, "Nnoise": 1,
           "Nmean": 7}
    rn, rg = va.gplite_nlZ(p["hyp"], gp0)
    assert nlz.shape == (1, 5) and g.shape == p["hyp"].shape
    same(nlz[0], rn); same(g, rg)
    (v,) = mex.call(1, "gp_nlz", p["hyp"], p["X"], p["y"].reshape(-1, 1), None, p["meanfun"], nf)
    same(v[0], va.gplite_nlZ(p["hyp"], gp0, nargout=1))
    rng = np.random.default_rng(4)
    a, b = np.asfortranarray(rng.standard_normal((3, 21))), np.asfortranarray(rng.standard_normal((3, 34)))
    same(mex.call(1, "sq_dist", a, b)[0], va.sq_dist(a, b))
    same(mex.call(1, "sq_dist", a)[0], va.sq_dist(a))
    same(mex.call(1, "sq_dist", a, None)[0], va.sq_dist(a))
    from tests._mex import MexError
    with pytest.raises(MexError) as e:
        mex.call(1, "sq_dist", a, np.zeros((2, 5)))
    assert e.value.identifier == "vbmc_hip:sq_dist"


def test_acquisition_commands(mex, va):
    """'acq', 'is_create', 'acq_iqr', 'is_free' as matlab/acqwrapper_vbmc.m / vbmc_hip_is_handle.m call them."""
    p, gp, vp, theta = make(seed=8, D=3, N=40, K=4, S=3)
    D = 3
    rng = np.random.default_rng(5)
    Xs = np.asfortranarray(rng.standard_normal((50, D)))
    gl = np.exp(0.1 * rng.standard_normal(D))
    gp2 = dict(gp, X_rescaled=gp["X"] / gl[None, :], sn2new=0.02 + 0.05 * rng.random(gp["X"].shape[0]))      # noise at the training points (N)
    Xa = np.asfortranarray(rng.standard_normal((24, D)))
    st = {"ymax": float(np.max(gp["y"])), "VarianceRegularizedAcqFcn": True, "TolGPVar": 1e-4, "gplengthscale": gl,
          "ActiveImportanceSampling": {"Xa": Xa}}
    hh = np.uint64(mex.call(1, "gp_upload", gp_struct(gp))[0][0, 0])
    for name, aid in (("acqf_vbmc", 0), ("acqflog_vbmc", 1), ("acqus_vbmc", 2), ("acqfsn2_vbmc", 3)):
        a, fb, vt = mex.call(3, "acq", hh, Xs, aid, vp_struct(vp), st["ymax"], 1, st["TolGPVar"], gl.reshape(1, -1), gp2["X_rescaled"],
                             gp2["sn2new"].reshape(-1, 1))
        ra, rfb, rvt = va.acqwrapper_vbmc(Xs, vp, gp2, st, False, name, nargout=3)
        same(a[:, 0], ra); same(fb[:, 0], rfb); same(vt[:, 0], rvt)
    his = mex.call(1, "is_create", hh, Xa, None, None, None)[0]
    hi = np.uint64(his[0, 0])
    a, fb, vt = mex.call(3, "acq_iqr", hh, hi, Xs, gl.reshape(1, -1), gp2["X_rescaled"], gp2["sn2new"].reshape(-1, 1), 1, st["TolGPVar"])
    ra, rfb, rvt = va.acqwrapper_vbmc(Xs, vp, gp2, st, False, "acqviqr_vbmc", nargout=3)
    same(a[:, 0], ra); same(fb[:, 0], rfb); same(vt[:, 0], rvt)
    mex.call(0, "is_free", hi)
    mex.call(0, "gp_free", hh)


def test_errors_cross_the_boundary_as_matlab_ids(mex):
    """VBMC_ERR_UNSUPPORTED -> 'vbmc_hip:unsupported' (what every shim catches to fall through); INVALID messages that start
    with a reference error id keep it (negelcbo_vbmc.m:22-23); everything the failed call created is released."""
    from tests._mex import MexError

    p, gp, vp, theta = make(seed=9)
    gs = gp_struct(gp)
    gs["meanfun"] = 6                                    # a mean function outside {0, 1, 4}
    with pytest.raises(MexError) as e:
        mex.call(1, "gp_upload", gs)
    assert e.value.identifier == "vbmc_hip:unsupported"
    hh = np.uint64(mex.call(1, "gp_upload", gp_struct(gp))[0][0, 0])
    with pytest.raises(MexError) as e:                   # beta ~= 0 with gradient needs compute_var == 2
        mex.call(2, "elbo", hh, theta.reshape(-1, 1), vp_struct(vp), 10, 1, 1, 0, 1.0, None, None, 1, 3)
    assert e.value.identifier == "negelcbo_vbmc:vargrad", e.value.identifier
    th = theta.copy(); th[0] = np.nan
    with pytest.raises(MexError) as e:
        mex.call(1, "elbo", hh, th.reshape(-1, 1), vp_struct(vp), 10, 0, 0, 0, 0, None, None, 1, 3)
    assert e.value.identifier == "vbmc_hip:error" and "non-finite" in e.value.message
    with pytest.raises(MexError) as e:                   # a handle that lost its class (a double) is refused, not dereferenced
        mex.call(1, "elbo", float(hh), theta.reshape(-1, 1), vp_struct(vp), 10, 0, 0, 0, 0, None, None, 1, 3)
    assert e.value.identifier == "vbmc_hip:usage"
    with pytest.raises(MexError) as e:
        mex.call(1, "no_such_command")
    assert e.value.identifier == "vbmc_hip:usage"
    mex.call(0, "gp_free", hh)
    assert mex.live_arrays() == 0


def test_multi_device_session_commands(mex, va):
    """'comm_open' (must precede every other command: here after `clear mex`), 'comm_size', 'gp_upload_all',
    'elbo_batch_multi', 'gp_free_all' as matlab/vbmc_hip_sieve.m / vbmc_hip_gp_handle.m use them; one device on this box, so
    the communicator has one rank -- the values must be those of 'elbo_batch'."""
    from tests._mex import MexError

    with pytest.raises(MexError) as e:
        mex.call(1, "comm_open", 1)                     # a context exists already
    assert e.value.identifier == "vbmc_hip:usage"
    with pytest.raises(MexError) as e:
        mex.call(1, "gp_upload_all", {"X": np.zeros((1, 1))})
    assert e.value.identifier == "vbmc_hip:usage"
    assert mex.call(1, "comm_size")[0][0, 0] == 1
    mex.close()                                          # clear mex: the mexAtExit handler destroys the context
    assert mex.call(1, "comm_open", 1)[0][0, 0] == 1
    assert mex.call(1, "comm_size")[0][0, 0] == 1
    p, gp, vp, theta = make(seed=10)
    rng = np.random.default_rng(7)
    Th = np.asfortranarray(theta[:, None] + 0.05 * rng.standard_normal((theta.size, 7)))
    tb = bounds(vp, theta, rng)
    tbp = {k: (np.asarray(v).reshape(-1) if k in ("lb", "ub") else v) for k, v in tb.items()}
    hs = mex.call(1, "gp_upload_all", gp_struct(gp))[0]
    assert hs.dtype == np.uint64 and hs.shape == (1, 1)
    F, dF, varG = mex.call(3, "elbo_batch_multi", hs, Th, vp_struct(vp), 30, 0, 1, 0, tb, 77)
    ref = va.negelcbo_batch(Th, 0, vp, gp, 30, False, 1, tbp, seed=77)
    same(F[0], ref["F"]); same(varG[0], ref["varG"])
    F, dF = mex.call(2, "elbo_batch_multi", hs, Th, vp_struct(vp), 30, 1, 0, 0, tb, 78)
    ref = va.negelcbo_batch(Th, 0, vp, gp, 30, True, 0, tbp, seed=78)
    same(F[0], ref["F"]); same(dF, ref["dF"])
    # the single-device commands run on device 0 of the session with the first handle
    (F1,) = mex.call(1, "elbo_batch", np.uint64(hs[0, 0]), Th, vp_struct(vp), 30, 0, 0, 0, tb, 78)
    same(F1[0], ref["F"])
    mex.call(0, "gp_free_all", hs)
    mex.close()
    mex.call(0, "open", 0)                               # leave a plain session behind for the fixture's teardown
