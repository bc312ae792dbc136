"""GPU parity: HIP negelcbo path (through the C ABI) vs the oracle on identical inputs.

Tolerances: fp64 end to end.  ELBO pieces are compared at rel 1e-10 (north_star asks 1e-6); the
looser 1e-9 on gradients covers summation-order differences over N / Ns terms.
"""
import numpy as np
import pytest

from oracle import vbmc_ref as R
from tests._cases import golden_cases, load_golden, synth_problem, theta_from_inputs, vp_from_inputs

pytestmark = pytest.mark.gpu

RT_VAL = 1e-10
RT_GRAD = 1e-9


from tests._cases import relerr  # noqa: E402  (relative to the largest reference entry; no floor)


@pytest.fixture(scope="module")
def va():
    import vbmc_amd

    return vbmc_amd


def problem(seed, D, N, K, S, **kw):
    p = synth_problem(seed, D, N, K, S, **kw)
    gp = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=p["meanfun"], noisefun=p["noisefun"], s2=p["s2"])
    vp = R.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    theta = np.concatenate([p["mu"].reshape(-1, order="F"), np.log(p["sigma"]), np.log(p["lam"]), p["eta"]])
    return p, gp, vp, theta


@pytest.mark.parametrize("path", golden_cases())
def test_golden_vectors(va, path):
    """HIP path vs the committed 50-digit mpmath vectors (entmc, entlb, per-sample log joint)."""
    inp, exp = load_golden(path)
    vp = vp_from_inputs(inp)
    gp = R.gplite_post(inp["hyp"], inp["X"], inp["y"], meanfun=inp["meanfun"])
    for s, post in enumerate(gp["post"]):
        post["alpha"] = np.array(exp["alpha"][s])
    theta = theta_from_inputs(inp)
    S = inp["S"]
    r = va.negelcbo_batch(theta, 0, vp, gp, 2 * inp["Mh"], True, 0, eps=inp["eps"])
    assert relerr(r["H"][0], exp["entmc_H"]) < 1e-11
    assert relerr(r["dH"][:, 0], exp["entmc_dH"]) < 1e-11
    assert relerr(r["G"][0], np.mean(exp["G_s"])) < 1e-11
    assert relerr(r["dG"][:, 0], np.mean(np.array(exp["dG_s"]), axis=0)) < 1e-11
    r0 = va.negelcbo_batch(theta, 0, vp, gp, 0, True, 0)
    assert relerr(r0["H"][0], exp["entlb_H"]) < 1e-11
    assert relerr(r0["dH"][:, 0], exp["entlb_dH"]) < 1e-10
    rs = va.negelcbo_batch(theta, 0, vp, gp, 0, False, 0, separate_K=True)
    assert relerr(rs["I_sk"][:, :, 0], np.array(exp["I_sk"])) < 1e-11
    assert S == rs["I_sk"].shape[0]


CONFIGS = [
    # name, D, N, K, S, Ns, target
    ("C1-rosenbrock", 2, 30, 2, 1, 100, "rosenbrock"),      # BASELINE configs[0]: D = 2, K = 2, Ns = 100, one hyper-sample, Rosenbrock target
    ("C2-student", 6, 200, 10, 8, 1000, "student"),
    ("small-odd", 3, 17, 5, 2, 37, "lumpy"),
    ("D10", 10, 120, 12, 4, 200, "lumpy"),
    ("D13-pad", 13, 60, 6, 2, 64, "lumpy"),
    ("D20", 20, 90, 7, 2, 64, "lumpy"),
    ("K70", 4, 40, 70, 2, 66, "lumpy"),
    # K mod 16 in 1..4 (K > 16): the last components run as a lane-layout tail beside K / 16 full k-tiles (entropy_mfma.h, TL)
    ("tail-K17-D3", 3, 30, 17, 2, 40, "lumpy"),
    ("tail-K20-D1", 1, 20, 20, 1, 50, "lumpy"),
    ("tail-K18-D10", 10, 40, 18, 2, 70, "lumpy"),
    ("tail-K33-D32", 32, 40, 33, 1, 20, "lumpy"),
    ("tail-K36-D13", 13, 40, 36, 2, 36, "lumpy"),
    ("tail-K34-D6", 6, 40, 34, 2, 130, "student"),
    ("tail-K49-D5", 5, 40, 49, 2, 34, "lumpy"),
    ("tail-K50-D10", 10, 60, 50, 2, 100, "lumpy"),
    ("tail-K51-D15", 15, 40, 51, 1, 38, "lumpy"),
    ("tail-K52-D20", 20, 60, 52, 2, 40, "lumpy"),
    # 5..8 tail components: two values per lane
    ("tail8-K21-D3", 3, 30, 21, 2, 40, "lumpy"),
    ("tail8-K24-D10", 10, 40, 24, 2, 50, "lumpy"),
    ("tail8-K38-D6", 6, 40, 38, 2, 66, "student"),
    ("tail8-K40-D20", 20, 50, 40, 1, 36, "lumpy"),
    ("tail8-K53-D10", 10, 40, 53, 2, 40, "lumpy"),
    ("tail8-K56-D13", 13, 40, 56, 1, 34, "lumpy"),
    ("tail8-K77-D5", 5, 40, 77, 2, 34, "lumpy"),
    ("tail8-K110-D20", 20, 50, 110, 1, 32, "lumpy"),
    ("tail8-K150-D4", 4, 30, 150, 1, 32, "lumpy"),
    # multi-wave workgroups whose LAST wave holds fewer components than the others: its tail lanes idle (K = 97: 49 + 48) or are
    # partly filled (K = 213 on four waves: 54 + 54 + 54 + 51)
    ("tail-K97-HV2", 4, 30, 97, 1, 32, "lumpy"),
    ("tail8-K213-HV4", 3, 30, 213, 1, 32, "lumpy"),
    # neighbours without a tail: K mod 16 = 0 and 9
    ("K48-D10", 10, 40, 48, 2, 40, "lumpy"),
    ("K57-D10", 10, 40, 57, 2, 40, "lumpy"),
    # round 3: K = 57..64 at D >= 31 run on TWO waves with two k-tiles each (ent_hv_small, abi_elbo.hip), the last wave possibly short
    # (K = 59: 30 + 29); one-wave kernels with four k-tiles from D = 15 on, and with three from D = 27 on, are built for ONE wave per
    # SIMD (VBMC_ENT_ONE_WAVE, entropy_mfma.h)
    ("hv2-K64-D32", 32, 36, 64, 1, 32, "lumpy"),
    ("hv2-K59-D31", 31, 36, 59, 1, 32, "lumpy"),
    ("w1-K54-D32", 32, 36, 54, 1, 32, "lumpy"),
    ("w1-K55-D24", 24, 40, 55, 1, 32, "lumpy"),
    ("w1-K64-D20", 20, 40, 64, 1, 34, "lumpy"),
    ("w1-K57-D28", 28, 40, 57, 1, 32, "lumpy"),
    ("w1-K59-D16", 16, 40, 59, 2, 36, "lumpy"),
    ("w1-K48-D28", 28, 40, 48, 1, 32, "lumpy"),
    # ... and so are the four-wave kernels with four k-tiles per wave from D = 23 on and with three at D >= 31 (K = 193..256)
    ("w1-hv4-K230-D24", 24, 30, 230, 1, 32, "lumpy"),
    ("w1-hv4-K200-D31", 31, 30, 200, 1, 32, "lumpy"),
]


@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_elbo_and_gradient_match_oracle(va, cfg):
    _, D, N, K, S, Ns, target = cfg
    p, gp, vp, theta = problem(1, D, N, K, S, target=target)
    Mh = (Ns + 1) // 2
    eps = np.random.default_rng(5).standard_normal((K, Mh, D))
    ref = R.negelcbo_vbmc(theta, 0, vp, gp, Ns, True, 0, eps=eps)
    F, dF, G, H, varF, dH = va.negelcbo_vbmc(theta, 0, vp, gp, Ns, 1, 0, nargout=6, eps=eps)
    assert relerr(G, ref["G"]) < RT_VAL
    assert relerr(H, ref["H"]) < RT_VAL
    assert relerr(F, ref["F"]) < RT_VAL
    assert relerr(dH, ref["dH"]) < RT_GRAD
    assert relerr(dF, ref["dF"]) < RT_GRAD
    # value-only call (the sieve's form) gives the same value
    (F2,) = va.negelcbo_vbmc(theta, 0, vp, gp, Ns, 0, 0, nargout=1, eps=eps)
    assert relerr(F2, ref["F"]) < RT_VAL


def test_deterministic_entropy_branch(va):
    """Ns == 0 -> entlb_vbmc (negelcbo_vbmc.m:104-110), the sieve's default."""
    for K in (1, 2, 9):
        p, gp, vp, theta = problem(2, 5, 50, K, 3)
        ref = R.negelcbo_vbmc(theta, 0, vp, gp, 0, True, 0)
        F, dF, G, H, _, dH = va.negelcbo_vbmc(theta, 0, vp, gp, 0, 1, 0, nargout=6)
        assert relerr(H, ref["H"]) < RT_VAL and relerr(G, ref["G"]) < RT_VAL
        assert relerr(dH, ref["dH"]) < RT_GRAD and relerr(dF, ref["dF"]) < RT_GRAD


@pytest.mark.parametrize("flags", [(1, 1, 1, 0), (1, 1, 0, 0), (0, 1, 1, 1), (1, 0, 0, 1), (0, 0, 0, 1)])
def test_optimize_flag_subsets(va, flags):
    """theta packing honours vp.optimize_* (negelcbo_vbmc.m:33-51); warm-up has no eta block."""
    p, gp, vp, _ = problem(3, 4, 40, 6, 2)
    for name, f in zip(("optimize_mu", "optimize_sigma", "optimize_lambda", "optimize_weights"), flags):
        vp[name] = bool(f)
    parts = []
    if flags[0]:
        parts.append(vp["mu"].reshape(-1, order="F") + 0.05)
    if flags[1]:
        parts.append(np.log(vp["sigma"]) - 0.1)
    if flags[2]:
        parts.append(np.log(vp["lambda"]) + 0.02)
    if flags[3]:
        parts.append(vp["eta"] + 0.1)
    theta = np.concatenate(parts)
    if flags == (0, 0, 0, 1):
        # weights-only branch (negelcbo_vbmc.m:54,78-88 -> gplogjoint_weights.m) reads the cached per-component
        # terms vp.stats.I_sk; the device path recomputes them from the unchanged mu/sigma/lambda: same value
        st = R.gplogjoint(vp, gp, (0, 0, 0, 0), True, True, 1, separate_K=True)
        vp["stats"] = {"I_sk": st["I_sk"], "J_sjk": st["J_sjk"]}
    eps = np.random.default_rng(7).standard_normal((6, 20, 4))
    ref = R.negelcbo_vbmc(theta, 0, vp, gp, 40, True, 0, eps=eps)
    F, dF = va.negelcbo_vbmc(theta, 0, vp, gp, 40, 1, 0, eps=eps)
    assert dF.shape == ref["dF"].shape
    assert relerr(F, ref["F"]) < RT_VAL and relerr(dF, ref["dF"]) < RT_GRAD


def test_soft_bounds_and_weight_penalty(va):
    p, gp, vp, theta = problem(4, 3, 30, 5, 2)
    opts = dict(TolLength=1e-6, TolWeight=1e-2, TolConLoss=0.01, WeightPenalty=0.1)
    vpb, tb = R.vpbounds(vp, gp, opts)
    th = theta.copy()
    th[0] += 9.0
    th[4] -= 7.0
    th[3 * 5 + 1] = 3.0     # a log-sigma far above the scale bound
    th[-1] = -9.0           # eta below its bound, weight below threshold
    th[-2] = 1.0            # eta above its upper bound 0
    eps = np.random.default_rng(8).standard_normal((5, 16, 3))
    ref = R.negelcbo_vbmc(th, 0, vpb, gp, 32, True, 0, thetabnd=tb, eps=eps)
    ref0 = R.negelcbo_vbmc(th, 0, vpb, gp, 32, True, 0, eps=eps)
    assert ref["F"] - ref0["F"] > 1.0  # the penalties are actually active
    F, dF = va.negelcbo_vbmc(th, 0, vpb, gp, 32, 1, 0, False, tb, eps=eps)
    assert relerr(F, ref["F"]) < RT_VAL and relerr(dF, ref["dF"]) < RT_GRAD


def test_batch_equals_singles_and_is_deterministic(va):
    p, gp, vp, theta = problem(5, 6, 80, 8, 4)
    R_ = 5
    rng = np.random.default_rng(3)
    thetas = theta[:, None] + 0.05 * rng.standard_normal((theta.size, R_))
    eps = rng.standard_normal((R_, 8, 25, 6))
    b1 = va.negelcbo_batch(thetas, 0, vp, gp, 50, True, 0, eps=eps)
    b2 = va.negelcbo_batch(thetas, 0, vp, gp, 50, True, 0, eps=eps)
    assert np.array_equal(b1["F"], b2["F"]) and np.array_equal(b1["dF"], b2["dF"])  # bit-identical reruns
    for r in range(R_):
        ref = R.negelcbo_vbmc(thetas[:, r], 0, vp, gp, 50, True, 0, eps=eps[r])
        assert relerr(b1["F"][r], ref["F"]) < RT_VAL
        assert relerr(b1["dF"][:, r], ref["dF"]) < RT_GRAD


@pytest.mark.parametrize("D,K,Ns,R_", [(10, 50, 2000, 64), (10, 50, 2010, 64), (6, 20, 200, 400), (3, 40, 1500, 300), (13, 33, 700, 300), (14, 56, 330, 300)])
def test_walking_entropy_launch_equals_the_chunk_grid(va, D, K, Ns, R_, monkeypatch):
    """Wide batches (two or more waves per wave slot in the chunk grid) run the WALKING launch of the matrix-core entropy kernel: one wave per
    slot walks its share of all (restart, component) pairs' tiles, a partial record per (wave, pair), variable record counts per pair in the
    reduction.  Same draws, same tile body: against the chunk grid (VBMC_ENT_WALK=0) only the order of summation over a pair's records
    differs.  Shapes: the headline class, a ragged last tile, many pairs per wave, one / two k-tiles, a component tail of eight."""
    p, gp, vp, theta = problem(40 + D, D, 30, K, 2)
    thetas = theta[:, None] + 0.05 * np.random.default_rng(D).standard_normal((theta.size, R_))
    monkeypatch.setenv("VBMC_ENT_KERNEL", "mfma")          # (small mixtures: not the lane kernel)
    monkeypatch.delenv("VBMC_ENT_WALK", raising=False)
    w1 = va.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, seed=77)
    w2 = va.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, seed=77)
    assert np.array_equal(w1["H"], w2["H"]) and np.array_equal(w1["dH"], w2["dH"])      # deterministic
    monkeypatch.setenv("VBMC_ENT_WALK", "0")
    c = va.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, seed=77)
    assert relerr(w1["H"], c["H"]) < 1e-13
    for r in range(R_):
        assert relerr(w1["dH"][:, r], c["dH"][:, r]) < 1e-12
    assert not np.array_equal(w1["dH"], c["dH"]), "the walking launch did not run (same bits as the chunk grid)"
    assert np.array_equal(w1["G"], c["G"])      # the log joint is untouched


@pytest.mark.parametrize("R_,Ns", [(1, 10000), (2, 10000), (1, 3000)])
def test_two_chunk_classes_where_the_launch_carries_the_role(va, R_, Ns, monkeypatch):
    """One or two restarts at a large sample count: the entropy launch carries the log-joint role, and the role's workgroups go on to a
    shorter chunk of the entropy (two chunk classes per component, EntArgs::co_c1 / co_c2 / co_tpc2).  Against the one-class chunk grid
    (VBMC_ENT_CHUNKS forces a uniform chunking) only the order of summation over a component's partial records differs."""
    p, gp, vp, theta = problem(61, 10, 80, 50, 20)
    thetas = theta[:, None] + 0.05 * np.random.default_rng(R_).standard_normal((theta.size, R_))
    monkeypatch.delenv("VBMC_ENT_CHUNKS", raising=False)
    a = va.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, seed=5)
    a2 = va.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, seed=5)
    assert np.array_equal(a["F"], a2["F"]) and np.array_equal(a["dF"], a2["dF"])
    monkeypatch.setenv("VBMC_ENT_CHUNKS", "12")
    b = va.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, seed=5)
    assert relerr(a["H"], b["H"]) < 1e-13 and relerr(a["G"], b["G"]) < 1e-13
    for r in range(R_):
        assert relerr(a["dH"][:, r], b["dH"][:, r]) < 1e-12 and relerr(a["dG"][:, r], b["dG"][:, r]) < 1e-12
    assert not np.array_equal(a["dH"], b["dH"])


def test_device_rng_stream_is_reproducible_and_standard_normal(va):
    """eps_mode 0: the Philox stream the kernel consumes can be dumped and fed to the oracle."""
    p, gp, vp, theta = problem(6, 5, 40, 4, 2)
    eng = va.default_engine()
    Ns, seed = 96, 1234
    eps = eng.ctx.rng_dump(5, 4, 2, Ns, seed)
    assert eps.shape == (2, 4, 48, 5)
    thetas = np.stack([theta, theta + 0.01], axis=1)
    out = va.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, seed=seed)
    for r in range(2):
        ref = R.negelcbo_vbmc(thetas[:, r], 0, vp, gp, Ns, True, 0, eps=eps[r])
        assert relerr(out["H"][r], ref["H"]) < RT_VAL
        assert relerr(out["dF"][:, r], ref["dF"]) < RT_GRAD
    big = eng.ctx.rng_dump(8, 16, 4, 4096, 99).reshape(-1)
    assert abs(big.mean()) < 5e-3 and abs(big.std() - 1.0) < 5e-3
    assert abs(np.mean(big**3)) < 2e-2 and abs(np.mean(big**4) - 3.0) < 5e-2
    out2 = va.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, seed=seed + 1)
    assert out2["H"][0] != out["H"][0]


@pytest.mark.parametrize("D", [2, 3, 4, 5, 7, 11, 15, 19])
def test_device_rng_stream_parity_in_every_padding_class(va, D):
    """The device-RNG tile carries draws in the padded dimensions and in the samples beyond Mh (round 4: no selects on the way into
    LDS): every D mod 4 -- D = 4 QS - 5 pads the whole last dim-block and a slot of the one before -- and sample counts whose
    last tile is partial, with and without the gradient, against the oracle on the dumped stream (ent/entmc_vbmc.m:49-104)."""
    K = 3
    p, gp, vp, theta = problem(40 + D, D, 30, K, 2)
    eng = va.default_engine()
    for Ns, seed in ((46, 5), (32, 6), (70, 7)):        # Mh = 23, 16, 35
        eps = eng.ctx.rng_dump(D, K, 1, Ns, seed)[0]
        ref = R.negelcbo_vbmc(theta, 0, vp, gp, Ns, True, 0, eps=eps)
        out = va.negelcbo_batch(theta[:, None], 0, vp, gp, Ns, True, 0, seed=seed)
        assert relerr(out["H"][0], ref["H"]) < RT_VAL
        assert relerr(out["dF"][:, 0], ref["dF"]) < RT_GRAD
        val = va.negelcbo_batch(theta[:, None], 0, vp, gp, Ns, False, 0, seed=seed)
        assert relerr(val["H"][0], ref["H"]) < RT_VAL


@pytest.mark.parametrize("D,K", [(5, 70), (7, 100), (6, 130), (9, 200), (3, 256)])
def test_device_rng_stream_parity_multi_wave_workgroups(va, D, K):
    """K > 64: two- and four-wave workgroups (wave 0 draws and stages the tile for all; round 4: the tile holds eps sigma_j there too and
    the antithetic pair shares the even part of the exponent, the second sign's exponents parked in LDS) on the device stream, partial
    last tile included, against the oracle on the dumped stream (ent/entmc_vbmc.m:49-104)."""
    p, gp, vp, theta = problem(60 + D, D, 25, K, 2)
    eng = va.default_engine()
    for Ns, seed in ((38, 3), (64, 4)):                 # Mh = 19 (partial tile), 32
        eps = eng.ctx.rng_dump(D, K, 1, Ns, seed)[0]
        ref = R.negelcbo_vbmc(theta, 0, vp, gp, Ns, True, 0, eps=eps)
        out = va.negelcbo_batch(theta[:, None], 0, vp, gp, Ns, True, 0, seed=seed)
        assert relerr(out["H"][0], ref["H"]) < RT_VAL
        assert relerr(out["dF"][:, 0], ref["dF"]) < RT_GRAD
        val = va.negelcbo_batch(theta[:, None], 0, vp, gp, Ns, False, 0, seed=seed)
        assert relerr(val["H"][0], ref["H"]) < RT_VAL


def test_entropy_mc_converges_to_closed_form(va):
    """K = 1: entmc -> 0.5 D (1 + log 2 pi) + D log sigma + sum log lambda (entlb_vbmc.m:34)."""
    p, gp, vp, theta = problem(7, 4, 30, 1, 1)
    (F0, _, _, H0) = va.negelcbo_vbmc(theta, 0, vp, gp, 0, 0, 0, nargout=4)
    (F1, _, _, H1) = va.negelcbo_vbmc(theta, 0, vp, gp, 200000, 0, 0, nargout=4, seed=5)
    assert abs(H1 - H0) < 0.02


def test_error_ids_mirror_reference(va):
    p, gp, vp, theta = problem(8, 3, 20, 3, 1)
    with pytest.raises(va.VbmcHipError, match="negelcbo_vbmc:vargrad"):
        va.negelcbo_vbmc(theta, 1.0, vp, gp, 10, 1, 1)
    gp_bad = dict(gp)
    gp_bad["meanfun"] = 6
    with pytest.raises(va.VbmcUnsupported, match="UnsupportedMeanFun"):
        va.negelcbo_vbmc(theta, 0, vp, gp_bad, 10, 1, 0)
    with pytest.raises(va.VbmcHipError, match="non-finite"):
        bad = theta.copy()
        bad[0] = np.nan
        va.negelcbo_vbmc(bad, 0, vp, gp, 10, 1, 0)


@pytest.mark.parametrize("variant", [0, 1])
def test_device_exp_accuracy(va, variant):
    """The hot-loop exp implementations: <= 2 ulp over the working range, saturation at the ends."""
    ctx = va.default_engine().ctx
    rng = np.random.default_rng(0)
    ends = [0.0, -0.0, 1.0, -1.0, 709.0, -745.0, -800.0, -1e5, -1e6, -2e6, -5e9, -1e300]
    x = np.concatenate([rng.uniform(-745, 709, 20000), rng.uniform(-40, 5, 20000), rng.uniform(-1e-3, 1e-3, 2000),
                        np.array(ends)])
    y = ctx.test_exp(x, variant)
    ref = np.exp(np.maximum(x, -1e4))
    ok = ref > 1e-300  # normal range
    rel = np.abs(y[ok] - ref[ok]) / ref[ok]
    assert rel.max() < 4.5e-16, rel.max()
    assert np.all(y[x <= -800] == 0.0) and np.all(y[~ok] < 1e-299)
    assert np.isinf(ctx.test_exp(np.array([710.0, 1e4]), variant)).all()


@pytest.mark.parametrize("variant,base,slope,poly", [(2, 1.5, 0.5, 0.0), (4, 2.0, 1.0, 0.0), (5, 2.0, 1.0, 1.62e-12), (3, 2.0, 1.0, 1.62e-12)])
def test_device_exp_sum_mode_accuracy(va, variant, base, slope, poly):
    """vb_exp_tab<1> (one-constant range reduction; the VALU fallback's exp) and vb_exp_tab1k (the MFMA entropy kernel's exp:
    pre-scaled argument, 1024-entry table; variant 4: economised degree-3 polynomial, variant 5: the economised quadratic the kernel is
    built with since round 5 -- its truncation error (c/2)^3/24 = 1.6e-12 is `poly`, checked PER VALUE here -- and variant 3: whichever of
    the two the library was compiled with): relative error bounded by poly + (base + slope |x|) ulp (the
    argument's own rounding: one rounded constant for the first, the rounded factor 1024/ln2 and the rounded product for the
    second), i.e. an absolute error below a few 1e-16 wherever exp(x) <= 1, and the same saturation -- including arguments far below the
    int32 range of the scaled exponent (-5e9 * 1477 saturates in v_cvt_i32_f64)."""
    ctx = va.default_engine().ctx
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-745, 709, 20000), rng.uniform(-40, 5, 20000), rng.uniform(-1e-3, 1e-3, 2000),
                        np.array([0.0, -0.0, 1.0, -1.0, 709.0, -745.0, -800.0, -1e5, -1e6, -5e9, -1e300])])
    y = ctx.test_exp(x, variant)
    ref = np.exp(np.maximum(x, -1e4))
    ok = ref > 1e-300
    rel = np.abs(y[ok] - ref[ok]) / ref[ok]
    assert np.all(rel <= poly + (base + slope * np.abs(x[ok])) * 2.220446049250313e-16), rel.max()
    neg = ok & (x <= 0)
    assert np.abs(y[neg] - ref[neg]).max() < poly + base * 1.2e-16
    if variant == 5:      # the bound is tight: the quadratic's error is really of that size (an exponential this cheap is not free)
        assert rel.max() > 1.0e-12
    assert np.all(y[x <= -800] == 0.0) and np.isinf(ctx.test_exp(np.array([710.0, 1e4]), variant)).all()


def test_block_sparse_mode_is_exact_to_rounding(va):
    """sparse_cutoff = 100 skips component tiles whose terms are < e^-100 of q: results unchanged to ~1e-14,
    on well-separated mixtures (most tiles skipped) and on overlapping ones (nothing skipped)."""
    for scale, K in ((1.0, 40), (0.02, 40), (1.0, 70)):
        p, gp, vp, theta = problem(51, 6, 60, K, 2)
        vp = dict(vp)
        vp["mu"] = vp["mu"] * scale          # scale << 1: all components overlap
        theta = theta.copy()
        theta[: 6 * K] *= scale
        Ns = 64
        eps = np.random.default_rng(3).standard_normal((K, Ns // 2, 6))
        d = va.negelcbo_batch(theta, 0, vp, gp, Ns, True, 0, eps=eps)
        s_ = va.negelcbo_batch(theta, 0, vp, gp, Ns, True, 0, eps=eps, sparse_cutoff=100.0)
        assert relerr(s_["H"], d["H"]) < 1e-13 and relerr(s_["dH"], d["dH"]) < 1e-12
        v = va.negelcbo_batch(theta, 0, vp, gp, Ns, False, 0, eps=eps, sparse_cutoff=100.0)
        assert relerr(v["H"], d["H"]) < 1e-13
        ref = R.negelcbo_vbmc(theta, 0, vp, gp, Ns, True, 0, eps=eps)
        assert relerr(s_["H"][0], ref["H"]) < RT_VAL and relerr(s_["dF"][:, 0], ref["dF"]) < RT_GRAD


SWEEP = [(1, 3), (7, 16), (14, 17), (15, 33), (18, 64), (22, 65), (23, 40), (30, 20), (32, 12), (5, 128), (3, 130), (9, 1),
         (26, 100), (12, 96), (10, 97), (32, 70),   # these four: components split over two waves (64 < K <= 128)
         (32, 140)]                                  # largest D with a K beyond the MFMA kernels: VALU fallback, big finalize record


@pytest.mark.parametrize("dk", SWEEP, ids=["D%dK%d" % t for t in SWEEP])
def test_kernel_instantiation_sweep(va, dk):
    """Every (QS, KT) family of the MFMA entropy kernel and the VALU fallback (K > 128), value and gradient."""
    D, K = dk
    p, gp, vp, theta = problem(61, D, 30, K, 2)
    Ns = 34
    eps = np.random.default_rng(6).standard_normal((K, Ns // 2, D))
    ref = R.negelcbo_vbmc(theta, 0, vp, gp, Ns, True, 0, eps=eps)
    F, dF, G, H, _, dH = va.negelcbo_vbmc(theta, 0, vp, gp, Ns, 1, 0, nargout=6, eps=eps)
    assert relerr(H, ref["H"]) < RT_VAL and relerr(G, ref["G"]) < RT_VAL
    assert relerr(dH, ref["dH"]) < RT_GRAD and relerr(dF, ref["dF"]) < RT_GRAD
    (Fv,) = va.negelcbo_vbmc(theta, 0, vp, gp, Ns, 0, 0, nargout=1, eps=eps)
    assert relerr(Fv, ref["F"]) < RT_VAL


def test_oversized_mixture_is_refused_cleanly(va):
    """Beyond the supported shapes the library answers with a clean VBMC_ERR_UNSUPPORTED (the shim falls through to the
    reference), not with a failed launch: more than 512 components (round 5; 256 through round 4).  (A D x K record too large for
    k_finalize's LDS, entlb beyond 128 components, the Monte-Carlo entropy of 128 < K <= 256 components and -- eight-wave workgroups --
    of 256 < K <= 512 are no longer refused: tests/test_gpu_limits.py.)"""
    p, gp, vp, theta = problem(5, 4, 40, 513, 1)
    with pytest.raises(va.VbmcUnsupported):
        va.negelcbo_vbmc(theta, 0, vp, gp, 0, 1, 0)
    with pytest.raises(va.VbmcUnsupported):
        va.negelcbo_vbmc(theta, 0, vp, gp, 20, 1, 0)
    # ... and 257 components, the old limit, evaluate
    p, gp, vp, theta = problem(5, 4, 40, 257, 1)
    eps = np.random.default_rng(2).standard_normal((257, 10, 4))
    ref = R.negelcbo_vbmc(theta, 0, vp, gp, 20, True, 0, eps=eps)
    F, dF = va.negelcbo_vbmc(theta, 0, vp, gp, 20, 1, 0, eps=eps)
    assert relerr(F, ref["F"]) < RT_VAL and relerr(dF, ref["dF"]) < RT_GRAD


def test_tiny_sigma_components_do_not_break_the_exp(va):
    """VBMC starts with sigma = 1e-3 (misc/setupvars_vbmc.m:83): exponents of -1e8 and below must give exactly 0."""
    p, gp, vp, theta = problem(62, 4, 30, 6, 1)
    th = theta.copy()
    th[4 * 6: 4 * 6 + 6] = np.log(1e-4)   # log sigma
    eps = np.random.default_rng(7).standard_normal((6, 16, 4))
    ref = R.negelcbo_vbmc(th, 0, vp, gp, 32, True, 0, eps=eps)
    F, dF, G, H = va.negelcbo_vbmc(th, 0, vp, gp, 32, 1, 0, nargout=4, eps=eps)
    assert np.isfinite(H) and relerr(H, ref["H"]) < RT_VAL and relerr(dF, ref["dF"]) < 1e-8


def test_output_subset_equals_full_outputs(va):
    """outputs=("F","dF") (the optimiser-loop call) moves only the leading part of each record; same numbers."""
    p, gp, vp, theta = problem(77, 5, 40, 6, 3)
    Th = np.asfortranarray(theta[:, None] + 0.01 * np.random.default_rng(0).standard_normal((theta.size, 7)))
    full = va.negelcbo_batch(Th, 0, vp, gp, 64, True, 0, seed=9)
    part = va.negelcbo_batch(Th, 0, vp, gp, 64, True, 0, seed=9, outputs=("F", "dF"))
    assert set(part) == {"F", "dF"}
    assert np.array_equal(part["F"], full["F"]) and np.array_equal(part["dF"], full["dF"])
    one = va.negelcbo_batch(Th[:, :1], 0, vp, gp, 64, True, 0, seed=9, outputs=("F", "dF"))
    assert np.array_equal(one["F"], full["F"][:1]) and np.array_equal(one["dF"], full["dF"][:, :1])


@pytest.mark.parametrize("shape", [(1, 1, 9, 1), (3, 16, 40, 2), (10, 50, 130, 3), (17, 33, 70, 2), (32, 20, 50, 2), (6, 100, 64, 2),
                                   (12, 7, 65, 1)])
def test_logjoint_mfma_and_valu_kernels_agree_with_oracle(va, shape, monkeypatch):
    """k_logjoint_mfma (gradient sums as moments on the matrix cores; chosen for large S x R grids) and k_logjoint
    (VALU sums) forced in turn on the same small problems: both against the oracle, all optimise-flag subsets."""
    D, K, N, S = shape
    p, gp, vp, theta = problem(500 + D + K, D, N, K, S)
    Th = np.asfortranarray(theta[:, None] + 0.02 * np.random.default_rng(1).standard_normal((theta.size, 3)))
    ref = [R.negelcbo_vbmc(Th[:, r], 0, vp, gp, 0, True, 0) for r in range(3)]
    for kern in ("mfma", "valu"):
        monkeypatch.setenv("VBMC_LJ_KERNEL", kern)
        out = va.negelcbo_batch(Th, 0, vp, gp, 0, True, 0)
        for r in range(3):
            assert relerr(out["G"][r], ref[r]["G"]) < 1e-10, (kern, shape)
            assert relerr(out["dG"][:, r], ref[r]["dG"]) < 1e-9, (kern, shape, relerr(out["dG"][:, r], ref[r]["dG"]))
    monkeypatch.setenv("VBMC_LJ_KERNEL", "mfma")
    for flags in [(1, 0, 0, 0), (0, 1, 1, 0), (1, 1, 1, 0)]:
        vpf = R.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"], optimize=flags)
        vpf["w"] = vp["w"]
        th, vpf = R.get_vptheta(vpf)
        o = va.negelcbo_batch(th, 0, vpf, gp, 0, True, 0)
        rr = R.negelcbo_vbmc(th, 0, vpf, gp, 0, True, 0)
        assert relerr(o["dG"][:, 0], rr["dG"]) < 1e-9, (flags, shape)


def test_prepared_objective_equals_batch_call(va):
    """PreparedObjective (arguments and buffers resolved once, as the closure of vpoptimize_vbmc.m:71) returns exactly what
    negelcbo_batch returns, call after call."""
    p, gp, vp, theta = problem(78, 5, 40, 6, 3)
    rng = np.random.default_rng(0)
    obj = va.PreparedObjective(theta.size, 4, 0, vp, gp, 40, 0, None)
    for it in range(3):
        Th = np.asfortranarray(theta[:, None] + 0.01 * rng.standard_normal((theta.size, 4)))
        F, dF = obj(Th, seed=100 + it)
        ref = va.negelcbo_batch(Th, 0, vp, gp, 40, True, 0, seed=100 + it)
        assert np.array_equal(F, ref["F"]) and np.array_equal(dF, ref["dF"])


def test_penalties_against_mpmath_vectors(va):
    """The soft-bound and weight penalties k_finalize adds (negelcbo_vbmc.m:136-164, vpbndloss.m, softbndloss.m) against the
    50-digit vectors tests/golden/mp_pen_case*.json -- every optimise-flag subset of the fixtures -- as the difference of the
    device evaluations with and without thetabnd (deterministic entropy: everything else cancels exactly)."""
    from tests._cases import load_pen_golden, pen_golden_cases
    from tests.test_oracle_golden import penalty_problem

    for path in pen_golden_cases():
        vp, theta, tb, exp = load_pen_golden(path)
        gp = penalty_problem(vp)
        F1, dF1 = va.negelcbo_vbmc(theta, 0, vp, gp, 0, 1, 0, False, tb)
        F0, dF0 = va.negelcbo_vbmc(theta, 0, vp, gp, 0, 1, 0)
        sc = max(1.0, abs(F1))
        assert abs((F1 - F0) - (exp["L_bnd"] + exp["L_w"])) < 1e-12 * sc, path
        assert np.max(np.abs((dF1 - dF0) - (exp["dL_bnd"] + exp["dL_w"]))) < 1e-12 * max(1.0, np.max(np.abs(dF1))), path


@pytest.mark.parametrize("K", [12, 18, 50, 70])
def test_extreme_exponents_saturate_cleanly(va, K):
    """Components with sigma = 1e-7 beside ordinary ones: the exponents of the other components at their samples (and of them at
    the others' samples) are ~ -1e13, far below the int32 range of the pre-scaled exponent (x 1024/ln2): the kernel's exp must
    return exactly 0 there (saturating conversion, device_math.h: vb_exp_tab1k) on every path -- full k-tiles, the component
    tail (K = 18: the tiny component 17 is a tail component; K = 50), two-wave workgroups (K = 70) -- and H, dH must still match
    the oracle."""
    D, N, S, Ns = 3, 30, 2, 40
    p, gp, vp, theta = problem(21, D, N, K, S)
    sig = np.array(vp["sigma"], dtype=np.float64).copy()
    tiny = [1, K - 1] if K > 2 else [0]
    sig[tiny] = 1e-7
    vp = dict(vp, sigma=sig)
    theta = theta.copy()
    theta[D * K + np.array(tiny)] = np.log(1e-7)
    eps = np.random.default_rng(9).standard_normal((K, Ns // 2, D))
    ref = R.negelcbo_vbmc(theta, 0, vp, gp, Ns, True, 0, eps=eps)
    F, dF, G, H, varF, dH = va.negelcbo_vbmc(theta, 0, vp, gp, Ns, 1, 0, nargout=6, eps=eps)
    assert np.isfinite(H) and np.all(np.isfinite(dH))
    assert relerr(H, ref["H"]) < RT_VAL
    assert relerr(dH, ref["dH"]) < RT_GRAD
