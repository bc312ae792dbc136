"""CPU: the plain-C restatement (CPU-baseline port) against the NumPy oracle and the mpmath vectors."""
import numpy as np
import pytest

from oracle import c_oracle
from oracle import vbmc_ref as R
from tests._cases import golden_cases, load_golden, synth_problem, theta_from_inputs, vp_from_inputs


@pytest.mark.parametrize("openmp", [False, True])
def test_c_port_matches_numpy_oracle(openmp):
    p = synth_problem(5, 6, 80, 7, 3)
    gp = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=4)
    vp = R.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
    theta = np.concatenate([p["mu"].reshape(-1, order="F"), np.log(p["sigma"]), np.log(p["lam"]), p["eta"]])
    eps = np.random.default_rng(0).standard_normal((7, 40, 6))
    ref = R.negelcbo_vbmc(theta, 0, vp, gp, 80, True, 0, eps=eps)
    alpha = np.stack([q["alpha"] for q in gp["post"]], axis=1)
    F, dF, G, H = c_oracle.negelcbo(theta, p["X"], p["hyp"], alpha, eps, openmp=openmp)
    assert abs(F - ref["F"]) < 1e-10 * max(1, abs(ref["F"]))
    assert abs(G - ref["G"]) < 1e-10 * max(1, abs(ref["G"])) and abs(H - ref["H"]) < 1e-10
    assert np.max(np.abs(dF - ref["dF"])) < 1e-9 * max(1, np.max(np.abs(ref["dF"])))


@pytest.mark.parametrize("path", golden_cases())
def test_c_port_matches_mpmath(path):
    inp, exp = load_golden(path)
    theta = theta_from_inputs(inp)
    alpha = np.array(exp["alpha"]).T
    F, dF, G, H = c_oracle.negelcbo(theta, inp["X"], inp["hyp"], alpha, inp["eps"], meanfun=inp["meanfun"])
    assert abs(H - exp["entmc_H"]) < 1e-11 * max(1, abs(exp["entmc_H"]))
    assert abs(G - np.mean(exp["G_s"])) < 1e-11 * max(1, abs(np.mean(exp["G_s"])))
    ref_dF = -np.mean(np.array(exp["dG_s"]), axis=0) - np.array(exp["entmc_dH"])
    assert np.max(np.abs(dF - ref_dF)) < 1e-10 * max(1, np.max(np.abs(ref_dF)))


def test_c_port_random_shapes():
    """The C port (serial and OpenMP builds) against the NumPy oracle over random small shapes -- the CPU leg of the three-way
    cross-check of SURVEY 8c(4) (the device leg is tests/test_gpu_random_shapes.py)."""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    @settings(max_examples=25, deadline=None, derandomize=True)
    @given(shape=st.tuples(st.integers(1, 7), st.integers(1, 9), st.integers(4, 40), st.integers(1, 3)), seed=st.integers(0, 10**6),
           mh=st.integers(1, 12), meanfun=st.sampled_from([0, 1, 4]), openmp=st.booleans())
    def check(shape, seed, mh, meanfun, openmp):
        D, K, N, S = shape
        p = synth_problem(seed, D, N, K, S, meanfun=meanfun)
        gp = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=meanfun)
        vp = R.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
        vp["w"] = np.exp(p["eta"]) / np.sum(np.exp(p["eta"]))
        theta = np.concatenate([p["mu"].reshape(-1, order="F"), np.log(p["sigma"]), np.log(p["lam"]), p["eta"]])
        eps = np.random.default_rng(seed + 1).standard_normal((K, mh, D))
        ref = R.negelcbo_vbmc(theta, 0, vp, gp, 2 * mh, True, 0, eps=eps)
        alpha = np.stack([q["alpha"] for q in gp["post"]], axis=1)
        F, dF, G, H = c_oracle.negelcbo(theta, p["X"], p["hyp"], alpha, eps, meanfun=meanfun, openmp=openmp)
        assert abs(G - ref["G"]) < 1e-10 * max(1, abs(ref["G"])) and abs(H - ref["H"]) < 1e-10 * max(1, abs(ref["H"]))
        assert np.max(np.abs(dF - ref["dF"])) < 1e-9 * max(1, np.max(np.abs(ref["dF"])))

    check()
