"""CPU: the host-side mirror (vbmc_amd/vp.py, optimize.py) against the oracle's restatement of the same
reference functions -- theta packing, rescaling, soft bounds, MATLAB sort semantics, Adam, vbinit shapes."""
import numpy as np
import pytest

import vbmc_amd.optimize as opt
import vbmc_amd.vp as vpm
from oracle import vbmc_ref as R
from tests._cases import synth_problem


def mk(seed=0, D=4, N=30, K=5, S=2, flags=(1, 1, 1, 1)):
    p = synth_problem(seed, D, N, K, S)
    vp = vpm.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"], optimize=flags)
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    gp = {"X": p["X"], "y": p["y"]}
    return p, vp, gp


def test_theta_packing_and_rescale_match_oracle():
    for flags in [(1, 1, 1, 1), (1, 1, 1, 0), (0, 1, 1, 1), (1, 0, 0, 1)]:
        p, vp, gp = mk(flags=flags)
        th_a, vp_a = vpm.get_vptheta(vp)
        th_b, vp_b = R.get_vptheta(vp)
        assert np.array_equal(th_a, th_b)
        for k in ("mu", "sigma", "lambda", "w"):
            assert np.array_equal(vp_a[k], vp_b[k])
        th2 = th_a + 0.1
        a, b = vpm.rescale_params(vp, th2), R.rescale_params(vp, th2)
        for k in ("mu", "sigma", "lambda", "w"):
            assert np.array_equal(a[k], b[k])
        assert ("eta" in a) == (not flags[3])   # eta removed only when weights are optimised (:36-39)


def test_vpbounds_match_oracle_and_accumulate():
    p, vp, gp = mk()
    o = dict(TolLength=1e-6, TolWeight=1e-2, TolConLoss=0.01, WeightPenalty=0.1)
    va, ta = vpm.vpbounds(vp, gp, o)
    vb, tb = R.vpbounds(vp, gp, o)
    for k in ("lb", "ub"):
        assert np.array_equal(ta[k], tb[k])
    assert ta["WeightThreshold"] == tb["WeightThreshold"] and ta["TolCon"] == tb["TolCon"]
    gp2 = {"X": p["X"] * 0.5, "y": p["y"]}
    va2, ta2 = vpm.vpbounds(va, gp2, o)  # bounds only widen (vpbounds.m:18-24)
    assert np.all(ta2["lb"] <= ta["lb"] + 1e-15) and np.all(ta2["ub"] >= ta["ub"] - 1e-15)


def test_matlab_sort_semantics():
    v = np.array([3.0, 1.0, np.nan, 1.0, -2.0, np.inf])
    assert list(opt.sort_ascend(v)) == list(R.matlab_sort_ascend(v)) == [4, 1, 3, 0, 5, 2]
    assert list(opt.sort_descend(v)) == list(R.matlab_sort_descend(v))
    X = np.arange(20.0).reshape(10, 2)
    y = np.array([1, 5, 3, 5, 2, 9, 0, 7, 7, 4.0])
    a, b = opt.gethpd_vbmc(X, y, 0.8), R.gethpd_vbmc(X, y, 0.8)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_fminadam_matches_oracle_on_a_noisy_quadratic():
    A = np.diag([1.0, 4.0, 9.0, 0.5])
    noise = np.random.default_rng(0).standard_normal((3000, 4)) * 0.01
    cnt = {"a": 0, "b": 0}

    def fa(x):
        cnt["a"] += 1
        return 0.5 * x @ A @ x, A @ x + noise[cnt["a"]]

    def fb(x):
        cnt["b"] += 1
        return 0.5 * x @ A @ x, A @ x + noise[cnt["b"]]

    x0 = np.array([1.0, -1.0, 0.5, 2.0])
    xa, f_a, xta, fta, ita = opt.fminadam(fa, x0, None, None, 1e-3, 2000)
    xb, f_b, xtb, ftb, itb = R.fminadam(fb, x0, TolFun=1e-3, MaxIter=2000)
    assert ita == itb and np.array_equal(xta, xtb) and np.array_equal(fta, ftb) and f_a == f_b


def test_vbinit_types_and_shapes():
    p, vp, gp = mk(K=4)
    Xs, ys = opt.gethpd_vbmc(p["X"], p["y"], 0.8)
    rng = np.random.default_rng(1)
    for t in (1, 2, 3):
        vs, ty = opt.vbinit_vbmc(t, 5, vp, 6, Xs, ys, rng)   # Knew > K: new components spawned for type 1
        assert len(vs) == 5 and np.all(ty == t)
        for v in vs:
            assert v["mu"].shape == (4, 6) and v["sigma"].shape == (6,) and v["lambda"].shape == (4,)
            assert abs(np.sum(v["w"]) - 1) < 1e-12 and np.all(v["sigma"] > 0)
    vs, _ = opt.vbinit_vbmc(1, 3, vp, 4, Xs, ys, rng)
    assert np.array_equal(vs[0]["mu"], vp["mu"])            # first type-1 candidate is the old vp verbatim (:59-61)


def test_vbmc_rnd_moments_and_balanced_split():
    """vbmc_rnd.m: mixture draws have the mixture's mean / per-dimension variance; the balanced variant places
    floor(w N) draws in every component."""
    import vbmc_amd.acq as acq

    rng = np.random.default_rng(0)
    D, K = 3, 4
    vp = {"D": D, "K": K, "mu": rng.standard_normal((D, K)), "sigma": np.array([0.3, 0.5, 0.2, 0.4]),
          "lambda": np.array([0.8, 1.0, 1.2]), "w": np.array([0.1, 0.2, 0.3, 0.4])}
    X, I = acq.vbmc_rnd(vp, 200000, False, rng=np.random.default_rng(1))
    mean = vp["mu"] @ vp["w"]
    var = (vp["w"][None, :] * ((vp["sigma"][None, :] * vp["lambda"][:, None]) ** 2 + vp["mu"] ** 2)).sum(1) - mean**2
    assert np.max(np.abs(X.mean(0) - mean)) < 0.01 and np.max(np.abs(X.var(0) - var) / var) < 0.02
    assert np.max(np.abs(np.bincount(I, minlength=K) / 200000 - vp["w"])) < 0.005
    Xb, Ib = acq.vbmc_rnd(vp, 1003, False, True, rng=np.random.default_rng(2))
    cnt = np.bincount(Ib, minlength=K)
    assert Xb.shape == (1003, D) and np.all(cnt >= np.floor(vp["w"] * 1003) - 3) and cnt.sum() == 1003
    X1, I1 = acq.vbmc_rnd(dict(vp, K=1, mu=vp["mu"][:, :1], sigma=vp["sigma"][:1], w=np.ones(1)), 50000, False, rng=np.random.default_rng(3))
    assert np.max(np.abs(X1.std(0) - vp["sigma"][0] * vp["lambda"])) < 0.01 and np.all(I1 == 0)


def test_grad_flag_defaulting_of_the_standalone_wrappers():
    """entmc_vbmc / entlb_vbmc / gplogjoint resolve grad_flags like the reference (ent/entmc_vbmc.m:5-11,
    misc/gplogjoint.m:17-23): no second output -> no gradient, omitted flags -> all four groups, a scalar flag is
    broadcast; theta then holds exactly the flagged groups, with or without the Jacobians.  Host logic only."""
    from vbmc_amd.elbo import _with_grad_groups
    rng = np.random.default_rng(0)
    D, K = 3, 4
    vp = vpm.make_vp(rng.standard_normal((D, K)), np.exp(rng.standard_normal(K)), np.ones(D), eta=rng.standard_normal(K))
    vp["w"] = np.exp(vp["eta"]) / np.sum(np.exp(vp["eta"]))
    _, th, g = _with_grad_groups(vp, None, 1, True, "entmc_vbmc")
    assert not g and th.size == D * K + K + D + K                  # value only: theta of the vp's own flags
    vpt, th, g = _with_grad_groups(vp, None, 2, True, "entmc_vbmc")
    assert g and all(vpt["optimize_" + n] for n in ("mu", "sigma", "lambda", "weights")) and th.size == D * K + K + D + K
    vpt, th, g = _with_grad_groups(vp, (0, 1, 0, 1), 2, True, "gplogjoint")
    assert g and (vpt["optimize_mu"], vpt["optimize_sigma"], vpt["optimize_lambda"], vpt["optimize_weights"]) == (False, True, False, True)
    assert th.size == 2 * K
    vpt, th, g = _with_grad_groups(vp, True, 2, True, "entlb_vbmc")
    assert g and th.size == D * K + K + D + K
    vpt, th, g = _with_grad_groups(vp, True, 2, False, "entlb_vbmc")   # JACOBIAN_FLAG = 0: same theta, the device drops the Jacobians
    assert g and th.size == D * K + K + D + K
    assert vp["optimize_mu"] and vp["optimize_weights"]            # the caller's vp is not modified


def test_copy_vp_is_independent_and_complete():
    """copy_vp stands in for MATLAB's value semantics (every `vp0_vec(i) = vp` of misc/vbinit_vbmc.m copies the struct): equal to
    copy.deepcopy field by field, and nothing of the copy aliases the original -- arrays, nested stats / bounds, lists."""
    import copy

    p, vp, gp = mk()
    vp["stats"] = {"I_sk": np.arange(10.0).reshape(2, 5), "J_sjk": np.ones((2, 5, 5)), "elbo": 1.5, "nested": {"a": np.zeros(3)}}
    vp["bounds"] = {"mu_lb": np.zeros(4), "mu_ub": np.ones(4)}
    vp["trinfo"] = None
    vp["hist"] = [np.zeros(2), {"b": np.ones(2)}]
    c, d = vpm.copy_vp(vp), copy.deepcopy(vp)

    def same(a, b):
        if isinstance(a, np.ndarray):
            return isinstance(b, np.ndarray) and a.shape == b.shape and np.array_equal(a, b)
        if isinstance(a, dict):
            return isinstance(b, dict) and a.keys() == b.keys() and all(same(a[k], b[k]) for k in a)
        if isinstance(a, list):
            return isinstance(b, list) and len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
        return a == b or (a is None and b is None)

    assert same(c, d)
    c["mu"][0, 0] += 1.0
    c["stats"]["I_sk"][0, 0] = -7.0
    c["stats"]["nested"]["a"][0] = 3.0
    c["bounds"]["mu_lb"][1] = 9.0
    c["hist"][1]["b"][0] = 5.0
    assert same(vp, d)          # the original saw none of it


def test_build_dependency_lists_cover_every_included_source():
    """vbmc_amd/build.py rebuilds a translation unit when a file of its dependency list is newer than the object: every local
    #include reachable from vbmc_hip.hip / ent_mfma_inst.hip has to be on the list, or an edited header ships a stale library."""
    import os
    import re

    from vbmc_amd import build as B

    def closure(start):
        seen, todo = set(), [start]
        while todo:
            f = todo.pop()
            if f in seen:
                continue
            seen.add(f)
            for inc in re.findall(r'^\s*#include\s+"([^"]+)"', open(os.path.join(B.CSRC, f)).read(), flags=re.M):
                if os.path.exists(os.path.join(B.CSRC, inc)):
                    todo.append(os.path.normpath(inc))
        return seen

    main = {os.path.normpath(d) for d in B.MAIN_DEPS}
    mfma = {os.path.normpath(d) for d in B.MFMA_DEPS}
    assert closure("vbmc_hip.hip") <= main, sorted(closure("vbmc_hip.hip") - main)
    assert closure("ent_mfma_inst.hip") <= mfma, sorted(closure("ent_mfma_inst.hip") - mfma)


def test_hardware_queue_default_is_set_by_the_library_and_never_overrides_the_environment():
    """Round 5: GPU_MAX_HW_QUEUES is raised to 8 inside vbmc_ctx_create, ahead of the library's first HIP call (busy
    streams that share a hardware queue run one after the other) -- the same for a ctypes, a MEX and a C host; a value already in the
    environment is kept, VBMC_HW_QUEUES=0 leaves the variable alone, and importing vbmc_amd no longer touches the environment."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys, ctypes as C; sys.path.insert(0, %r); import vbmc_amd._lib as L; a = os.environ.get('GPU_MAX_HW_QUEUES');"
            "lib = L.load(); h = C.c_void_p(); lib.vbmc_ctx_create(0, None, C.byref(h));"
            "libc = C.CDLL(None); libc.getenv.restype = C.c_char_p; v = libc.getenv(b'GPU_MAX_HW_QUEUES');"
            "print(a, v.decode() if v else None)") % root
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "VBMC_HW_QUEUES")}

    def run(e):
        return subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True).stdout.strip()

    assert run(env) == "None 8"
    assert run(dict(env, GPU_MAX_HW_QUEUES="3")) == "3 3"
    assert run(dict(env, VBMC_HW_QUEUES="0")) == "None None"
    assert run(dict(env, VBMC_HW_QUEUES="6")) == "None 6"
