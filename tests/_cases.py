"""Shared helpers for the tests: load golden cases, build seeded synthetic problems."""
import glob
import json
import os

import numpy as np

from oracle import vbmc_ref as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    return sorted(glob.glob(os.path.join(GOLDEN, "mp_case*.json")))


def nlz_golden_cases():
    return sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "mp_nlz_case*.json")))


def load_nlz_golden(path):
    """-> (gp dict without posterior, hyp Nhyp x S, expected dict)."""
    with open(path) as f:
        rec = json.load(f)
    inp = rec["inputs"]
    X = np.array(inp["X"], dtype=np.float64)
    D = X.shape[1]
    nf = tuple(inp["noisefun"])
    gp = {"X": X, "y": np.array(inp["y"], dtype=np.float64),
          "s2": None if inp["s2"] is None else np.array(inp["s2"], dtype=np.float64),
          "covfun": 1, "Ncov": D + 1, "noisefun": nf, "Nnoise": R.noisefun_nhyp(nf), "meanfun": inp["meanfun"],
          "Nmean": R.meanfun_nhyp(inp["meanfun"], D), "meanfun_extras": None, "intmeanfun": 0}
    return gp, np.array(inp["hyp"], dtype=np.float64), {k: np.array(v, dtype=np.float64) for k, v in rec["expected"].items()}


def pred_golden_cases():
    return sorted(glob.glob(os.path.join(GOLDEN, "mp_pred_case*.json")))


def load_pred_golden(path):
    """-> (inputs dict with arrays / None, expected dict of S x ... arrays): prediction with the general noise models."""
    with open(path) as f:
        rec = json.load(f)
    inp = {k: (np.array(v, dtype=np.float64) if isinstance(v, list) and k != "noisefun" else v) for k, v in rec["inputs"].items()}
    inp["noisefun"] = tuple(inp["noisefun"])
    return inp, {k: np.array(v, dtype=np.float64) for k, v in rec["expected"].items()}


def pen_golden_cases():
    return sorted(glob.glob(os.path.join(GOLDEN, "mp_pen_case*.json")))


def load_pen_golden(path):
    """-> (vp with the optimise flags of the case, theta, thetabnd, expected dict): soft-bound and weight penalties."""
    with open(path) as f:
        rec = json.load(f)
    inp = rec["inputs"]
    a = lambda k: np.array(inp[k], dtype=np.float64)  # noqa: E731
    vp = R.make_vp(a("mu"), a("sigma"), a("lam"), eta=a("eta"))
    e = np.exp(a("eta"))
    vp["w"] = e / np.sum(e)
    for name, o in zip(("optimize_mu", "optimize_sigma", "optimize_lambda", "optimize_weights"), inp["opt"]):
        vp[name] = bool(o)
    tb = {"lb": a("lb"), "ub": a("ub"), "TolCon": inp["TolCon"], "WeightThreshold": inp["WeightThreshold"], "WeightPenalty": inp["WeightPenalty"]}
    return vp, a("theta"), tb, {k: np.array(v, dtype=np.float64) for k, v in rec["expected"].items()}


def acq_golden_cases():
    return sorted(glob.glob(os.path.join(GOLDEN, "mp_acq_case*.json")))


def load_acq_golden(path):
    """-> (vp, gp with X_rescaled / sn2new, Xs, optimState with the importance points Xa only, expected dict)."""
    with open(path) as f:
        rec = json.load(f)
    inp = {k: (np.array(v, dtype=np.float64) if isinstance(v, list) else v) for k, v in rec["inputs"].items()}
    vp = R.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
    vp["w"] = inp["w"]
    gp = R.gplite_post(inp["hyp"], inp["X"], inp["y"], meanfun=inp["meanfun"])
    gp = dict(gp, X_rescaled=inp["X"] / inp["gplengthscale"][None, :], sn2new=inp["sn2new"])
    st = {"ymax": inp["ymax"], "VarianceRegularizedAcqFcn": False, "TolGPVar": 1e-4, "gplengthscale": inp["gplengthscale"],
          "ActiveImportanceSampling": {"Xa": inp["Xa"]}}
    exp = {k: np.array(v, dtype=np.float64) for k, v in rec["expected"].items()}
    return vp, gp, inp["Xstar"], st, exp


def load_golden(path):
    with open(path) as f:
        rec = json.load(f)
    inp = {k: (np.array(v, dtype=np.float64) if isinstance(v, list) else v) for k, v in rec["inputs"].items()}
    return inp, rec["expected"]


def vp_from_inputs(inp):
    vp = R.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
    e = np.exp(inp["eta"])
    vp["w"] = e / np.sum(e)  # negelcbo_vbmc.m:45-47
    return vp


def theta_from_inputs(inp):
    return np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])


def synth_problem(seed, D, N, K, S, meanfun=4, noisy=False, target="lumpy"):
    """Seeded synthetic GP + VP in the shape of SURVEY.md section 8(d)."""
    rng = np.random.default_rng(seed)
    X = 1.5 * rng.standard_normal((N, D))
    if target == "lumpy":
        nc = 12
        mus = rng.uniform(-2, 2, size=(nc, D))
        sig = rng.uniform(0.3, 1.0, size=nc)
        wts = rng.dirichlet(np.ones(nc))
        lp = np.stack([
            np.log(wts[i]) - 0.5 * np.sum(((X - mus[i]) / sig[i]) ** 2, axis=1) - D * np.log(sig[i]) - 0.5 * D * np.log(2 * np.pi)
            for i in range(nc)
        ])
        mx = lp.max(axis=0)
        y = mx + np.log(np.sum(np.exp(lp - mx), axis=0))
    elif target == "rosenbrock":   # BASELINE configs[0]: the reference's own test target (rosenbrock_test.m:7, noise-free form)
        y = -np.sum((X[:, :-1] ** 2 - X[:, 1:]) ** 2 + (X[:, :-1] - 1.0) ** 2 / 100.0, axis=1)
    else:  # multivariate Student-t, nu = 5, scale diag(1:D)/D
        nu = 5.0
        sc = np.arange(1, D + 1) / D
        y = -0.5 * (nu + D) * np.log1p(np.sum((X / sc) ** 2, axis=1) / nu)
    s2 = None
    noisefun = (1, 0, 0)
    if noisy:
        y = y + rng.standard_normal(N)
        s2 = np.ones(N)
        noisefun = (1, 1, 0)
    nmean = {0: 0, 1: 1, 4: 2 * D + 1}[meanfun]
    hyp = np.zeros((D + 2 + nmean, S))
    for s in range(S):
        hyp[:D, s] = np.log(0.8) + 0.2 * rng.standard_normal(D)
        hyp[D, s] = np.log(np.std(y)) + 0.1 * rng.standard_normal()
        hyp[D + 1, s] = np.log(1e-3)
        if meanfun >= 1:
            hyp[D + 2, s] = np.max(y)
        if meanfun == 4:
            hyp[D + 3 : D + 3 + D, s] = 0.2 * rng.standard_normal(D)
            hyp[D + 3 + D :, s] = np.log(2.0) + 0.1 * rng.standard_normal(D)
    order = np.argsort(-y, kind="stable")
    hpd = X[order[: int(round(0.8 * N))]]
    mu = hpd[rng.permutation(hpd.shape[0])[np.arange(K) % hpd.shape[0]]].T.copy()
    V = np.var(mu, axis=1, ddof=1) if K > 1 else np.var(hpd, axis=0, ddof=1)
    sigma = np.sqrt(np.mean(V) / K) * np.exp(0.2 * rng.standard_normal(K))
    lam = np.std(hpd, axis=0, ddof=1)
    lam = lam * np.sqrt(D / np.sum(lam**2))
    eta = 0.3 * rng.standard_normal(K)
    return dict(D=D, N=N, K=K, S=S, X=X, y=y, s2=s2, hyp=hyp, meanfun=meanfun, noisefun=noisefun,
                mu=mu, sigma=sigma, lam=lam, eta=eta, rng=rng)


def relerr(a, b, scale=None):
    """max |a - b| relative to the LARGEST reference entry (round 5: no floor at 1 -- a quantity below 1 used to get an absolute test).
    `scale`: an explicit yardstick for quantities that are differences of larger terms (say so at the call site); an all-zero reference
    is compared absolutely."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if not a.size:
        return 0.0
    den = float(np.max(np.abs(b))) if scale is None else float(scale)
    return float(np.max(np.abs(a - b)) / den) if den > 0.0 else float(np.max(np.abs(a - b)))


def block_relerr(a, b, D, K, opt=(1, 1, 1, 1)):
    """Per-block relative error of a theta-shaped gradient [mu (D K) | log sigma (K) | log lambda (D) | eta (K)] (only the
    optimised groups present): each block against ITS OWN largest reference entry, so that a wrong small block (a lambda
    gradient of 1e-4 beside a mu gradient of 1e2) cannot hide behind the largest entry of the whole vector.  A block that
    is itself pure cancellation noise (the mu-gradient of the entropy of well separated components: O(1) sample terms
    summing to 1e-9) is measured against 1e-6 of the vector's largest entry instead -- six orders below it, still far
    finer than the whole-vector check."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    sizes = [D * K if opt[0] else 0, K if opt[1] else 0, D if opt[2] else 0, K if opt[3] else 0]
    assert a.size == b.size == sum(sizes), (a.size, b.size, sizes)
    out, i = {}, 0
    floor = 1e-6 * float(np.max(np.abs(b))) if b.size else 0.0
    for name, n in zip(("mu", "sigma", "lambda", "eta"), sizes):
        if n:
            sc = max(float(np.max(np.abs(b[i:i + n]))), floor)
            out[name] = float(np.max(np.abs(a[i:i + n] - b[i:i + n]))) / sc if sc > 0 else float(np.max(np.abs(a[i:i + n])))
        i += n
    return out
