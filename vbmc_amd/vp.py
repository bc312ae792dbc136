"""Variational-posterior plumbing on the host (O(T) bookkeeping, no numerics on the path).

Mirrors misc/get_vptheta.m, misc/rescale_params.m, misc/vpbounds.m of the reference with the
same names and argument meaning.  A ``vp`` is a plain dict with the reference's field names
(D, K, mu[D,K], sigma[K], lambda[D], w[K], eta[K], optimize_mu/sigma/lambda/weights, delta,
bounds, stats) so that tests read like the reference's own code.
"""
from __future__ import annotations

import copy
import math

import numpy as np


def make_vp(mu, sigma, lambda_, w=None, eta=None, optimize=(True, True, True, True), delta=None):
    """Variational posterior struct (misc/setupvars_vbmc.m:78-99)."""
    mu = np.array(mu, dtype=np.float64)
    D, K = mu.shape
    vp = {
        "D": D, "K": K, "mu": mu,
        "sigma": np.array(sigma, dtype=np.float64).reshape(K),
        "lambda": np.array(lambda_, dtype=np.float64).reshape(D),
        "w": np.full(K, 1.0 / K) if w is None else np.array(w, dtype=np.float64).reshape(K),
        "optimize_mu": bool(optimize[0]), "optimize_sigma": bool(optimize[1]),
        "optimize_lambda": bool(optimize[2]), "optimize_weights": bool(optimize[3]),
        "delta": delta, "bounds": None, "stats": None,
    }
    if eta is not None:
        vp["eta"] = np.array(eta, dtype=np.float64).reshape(K)
    return vp


def copy_vp(vp):
    """An independent copy of a variational-posterior dict (what MATLAB's value semantics give the reference for free): arrays
    copied, nested dicts (stats, bounds, trinfo) one level deep with their arrays copied -- a tenth of copy.deepcopy's time,
    which was a fifth of a vpsieve_vbmc call with 100 candidates."""
    out = {}
    for k, v in vp.items():
        if isinstance(v, np.ndarray):
            out[k] = v.copy()
        elif isinstance(v, dict):
            out[k] = {kk: (vv.copy() if isinstance(vv, np.ndarray) else copy.deepcopy(vv) if isinstance(vv, (dict, list)) else vv)
                      for kk, vv in v.items()}
        elif isinstance(v, list):
            out[k] = copy.deepcopy(v)
        else:
            out[k] = v
    return out


def rescale_params(vp, theta=None):
    """misc/rescale_params.m:1-40: assign theta, renormalise lambda (sum lambda^2 = D), weights."""
    vp = copy_vp(vp)
    D = vp["D"]
    if theta is not None and np.size(theta) > 0:
        theta = np.asarray(theta, dtype=np.float64).reshape(-1)
        K = vp["K"]
        i0 = 0
        if vp["optimize_mu"]:
            vp["mu"] = theta[: D * K].reshape(D, K, order="F").copy()
            i0 = D * K
        if vp["optimize_sigma"]:
            vp["sigma"] = np.exp(theta[i0 : i0 + K])
            i0 += K
        if vp["optimize_lambda"]:
            vp["lambda"] = np.exp(theta[i0 : i0 + D])
        if vp["optimize_weights"]:
            eta = theta[-K:]
            vp["w"] = np.exp(eta - np.max(eta))
    nl = math.sqrt(float(np.sum(vp["lambda"] ** 2)) / D)
    vp["lambda"] = vp["lambda"] / nl
    vp["sigma"] = vp["sigma"] * nl
    if vp["optimize_weights"]:
        vp["w"] = vp["w"] / np.sum(vp["w"])
        vp.pop("eta", None)
    vp.pop("mode", None)
    return vp


def get_vptheta(vp, optimize_mu=None, optimize_sigma=None, optimize_lambda=None, optimize_weights=None):
    """misc/get_vptheta.m:1-22 -> (theta, vp)."""
    om = vp["optimize_mu"] if optimize_mu is None else optimize_mu
    os_ = vp["optimize_sigma"] if optimize_sigma is None else optimize_sigma
    ol = vp["optimize_lambda"] if optimize_lambda is None else optimize_lambda
    ow = vp["optimize_weights"] if optimize_weights is None else optimize_weights
    vp = rescale_params(vp)
    parts = []
    if om:
        parts.append(vp["mu"].reshape(-1, order="F"))
    if os_:
        parts.append(np.log(vp["sigma"]))
    if ol:
        parts.append(np.log(vp["lambda"]))
    if ow:
        parts.append(np.log(vp["w"]))
    return (np.concatenate(parts) if parts else np.zeros(0)), vp


def vpbounds(vp, gp, options, K=None):
    """misc/vpbounds.m:1-55 -> (vp, thetabnd).  Bounds accumulate in vp['bounds'] (only widen)."""
    K = vp["K"] if K is None else K
    D = vp["D"]
    vp = copy.deepcopy(vp)
    b = vp.get("bounds") or {
        "mu_lb": np.full(D, np.inf), "mu_ub": np.full(D, -np.inf),
        "lnscale_lb": np.full(D, np.inf), "lnscale_ub": np.full(D, -np.inf),
    }
    X = np.asarray(gp["X"], dtype=np.float64)
    xmin, xmax = X.min(axis=0), X.max(axis=0)
    b["mu_lb"] = np.minimum(xmin, b["mu_lb"])
    b["mu_ub"] = np.maximum(xmax, b["mu_ub"])
    lnrange = np.log(xmax - xmin)
    b["lnscale_lb"] = np.minimum(b["lnscale_lb"], lnrange + math.log(options["TolLength"]))
    b["lnscale_ub"] = np.maximum(b["lnscale_ub"], lnrange)
    if vp["optimize_weights"]:
        b["eta_lb"] = math.log(0.5 * options["TolWeight"])
        b["eta_ub"] = 0.0
    vp["bounds"] = b
    lb, ub = [], []
    if vp["optimize_mu"]:
        lb.append(np.tile(b["mu_lb"], K))
        ub.append(np.tile(b["mu_ub"], K))
    if vp["optimize_sigma"] or vp["optimize_lambda"]:
        lb.append(np.tile(b["lnscale_lb"], K))
        ub.append(np.tile(b["lnscale_ub"], K))
    if vp["optimize_weights"]:
        lb.append(np.full(K, b["eta_lb"]))
        ub.append(np.full(K, b["eta_ub"]))
    thetabnd = {"lb": np.concatenate(lb) if lb else np.zeros(0), "ub": np.concatenate(ub) if ub else np.zeros(0),
                "TolCon": options["TolConLoss"]}
    if vp["optimize_weights"]:
        thetabnd["WeightThreshold"] = max(1.0 / (4 * K), options["TolWeight"])
        thetabnd["WeightPenalty"] = options["WeightPenalty"]
    return vp, thetabnd


# VBMC defaults touching the path (vbmc.m:158-366)
DEFAULT_OPTIONS = {
    "TolLength": 1e-6, "TolWeight": 1e-2, "TolConLoss": 0.01, "WeightPenalty": 0.1, "HPDFrac": 0.8,
    "NSent": lambda K: 100 * K ** (2.0 / 3.0), "NSentFast": 0, "NSentFine": lambda K: 2**12 * K,
    "NSelbo": lambda K: 50 * K, "NSelboIncr": 0.1, "ElboStarts": 2, "ELCBOWeight": 0,
    "SGDStepSize": 0.005, "TolFunStochastic": 1e-3, "MaxIterStochastic": None, "ELCBOmidpoint": True,
    "StochasticOptimizer": "adam", "TolImprovement": 0.01, "ELCBOImproWeight": 3,
    "PruningThresholdMultiplier": lambda K: 1.0 / math.sqrt(K), "VariationalInitRepo": False,
    "DetEntTolOpt": 1e-3,
}


def evaloption(option, N):
    """misc/evaloption_vbmc.m:4-8."""
    return option(N) if callable(option) else option
