"""Shared by the CPU and GPU tests of vpoptimize_vbmc: the problem, the replay of the product's device-stream schedule
through the oracle (``eps_for`` built from the product's ``trace``), and the comparison of the two results."""
import numpy as np

from oracle import vbmc_ref as R
from tests._cases import synth_problem


def vpopt_problem(seed=41, D=3, N=40, K=4, S=2, weights=(0.55, 0.40, 0.03, 0.02)):
    p = synth_problem(seed, D, N, K, S)
    gp = R.gplite_post(p["hyp"], p["X"], p["y"], meanfun=4)
    vp = R.make_vp(p["mu"], p["sigma"], p["lam"], eta=p["eta"])
    vp["w"] = np.array(weights, dtype=np.float64)[:K] / np.sum(weights[:K])
    vp.pop("eta", None)
    return p, gp, vp


OPTS = {"MaxIterStochastic": 60, "NSentFine": lambda K: 96 * K, "NSent": lambda K: 24 * K,
        # thresholds chosen so that the pruning loop really runs on a 4-component toy mixture: several components are below
        # TolWeight, some prunings are accepted and some rejected
        "TolWeight": 0.3, "TolImprovement": 0.6}


def eps_from_trace(trace, stream):
    """eps_for(kind, slot, it, K, Ns) for the oracle: the draws the product's evaluation (kind, slot) consumed.
    ``stream(seed, r, R, K, Ns)`` -> (K, Ns/2, D) standard normals of restart r in a batch of R keyed by seed."""
    by = {(t["kind"], t["slot"]): t for t in trace if t["kind"] in ("adam", "full", "prune")}

    def eps_for(kind, slot, it, K, Ns):
        t = by[(kind, slot)]
        assert t["K"] == K and (Ns is None or t["Ns"] == Ns)
        return stream(t["seed"] + it, t["r"], t["R"], K, t["Ns"])

    return eps_for


def compare(vpa, vpb, varss_a, varss_b, pruned_a, pruned_b, tol):
    assert pruned_a == pruned_b and vpa["K"] == vpb["K"]
    for f in ("mu", "sigma", "lambda", "w"):
        assert np.allclose(vpa[f], vpb[f], rtol=tol, atol=tol), f
    sa, sb = vpa["stats"], vpb["stats"]
    for f in ("elbo", "elbo_sd", "elogjoint", "elogjoint_sd", "entropy"):
        assert abs(sa[f] - sb[f]) <= tol * max(1.0, abs(sb[f])), (f, sa[f], sb[f])
    assert sa["I_sk"].shape == sb["I_sk"].shape and sa["J_sjk"].shape == sb["J_sjk"].shape
    assert np.allclose(sa["I_sk"], sb["I_sk"], rtol=tol, atol=tol * max(1.0, np.max(np.abs(sb["I_sk"]))))
    assert np.allclose(sa["J_sjk"], sb["J_sjk"], rtol=tol, atol=tol * max(1.0, np.max(np.abs(sb["J_sjk"]))))
    assert abs(varss_a - varss_b) <= tol * max(1.0, abs(varss_b))
