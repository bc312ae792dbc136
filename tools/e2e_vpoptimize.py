"""What a VBMC user gets: one vpoptimize_vbmc call (misc/vpoptimize_vbmc.m:1-254 -- sieve over Nfastopts candidates, Nslowopts Adam
chains, full-variance ELCBO evaluations, pruning) with VBMC's own defaults at the BASELINE shape D = 10, N = 400, K = 50, S = 20,
on the device through vbmc_amd (the mirror of the reference call surface), with the evaluations it makes counted from its trace;
beside it the NumPy oracle's cost per evaluation of the same kinds on the box's host (the closest analogue here of MATLAB's
vectorised code), from which the same call's host time is projected.

    python tools/e2e_vpoptimize.py [out.md]          (GPU box; default gpurun_out/e2e_vpoptimize.md)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402
from oracle import vbmc_ref as R  # noqa: E402  (timed as the host baseline only)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "e2e_vpoptimize.md")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    D, N, K, S = 10, 400, 50, 20
    inp = synth_inputs(0, D, N, K, S)
    eng = vbmc_amd.Engine(0)
    gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng)
    vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
    vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
    vp.pop("eta", None)
    L = []
    P = L.append
    P("# One `vpoptimize_vbmc` call with VBMC's defaults at D = 10, N = 400, K = 50, S = 20 (tools/e2e_vpoptimize.py)\n")
    P("`vbmc_amd.vpoptimize_vbmc(Nfastopts, Nslowopts = 2, vp, gp, K)` -- the mirror of `misc/vpoptimize_vbmc.m` -- on one MI355X, options "
      "as `vbmc.m` sets them (`NSent = 100 K^(2/3)`, `NSentFine = 2^12 K`, `MaxIterStochastic = 100 (2 + D)`, `TolFunStochastic = 1e-3`, "
      "deterministic-entropy sieve).  Evaluations counted from the call's own trace.\n")
    P("| Nfastopts | wall s (device) | sieve evals | Adam iterations (2 chains) | full-ELCBO evals | pruned | ELBO |\n|---|---|---|---|---|---|---|")
    rows = []
    for nfast in (100, 10):        # full refit / incremental (vbmc.m:224-225,699,706 with the warm-up K)
        for rep in range(2):       # the second call runs with everything warm
            trace = []
            t0 = time.perf_counter()
            vpo, varss, pruned = vbmc_amd.vpoptimize_vbmc(nfast, 2, dict(vp), gp, K, {}, {}, seed=3 + rep, engine=eng, trace=trace,
                                                          rng=np.random.default_rng(rep))
            dt = time.perf_counter() - t0
        kinds = {}
        for t in trace:
            kinds.setdefault(t["kind"], []).append(t)
        n_adam = sum(int(t.get("iters", 0)) for t in kinds.get("adam", []))
        n_full = len(kinds.get("full", [])) + len(kinds.get("prune", []))     # one trace entry per evaluated vp
        n_sieve = nfast                                                         # the sieve evaluates its Nfastopts candidates once
        rows.append((nfast, dt, n_sieve, n_adam, n_full, pruned, vpo["stats"]["elbo"]))
        P("| %d | %.3f | %d | %d | %d | %d | %.4f |" % rows[-1])
    # ---- host cost per evaluation, NumPy oracle (same kinds of calls)
    gpo = R.gplite_post(inp["hyp"], inp["X"], inp["y"], meanfun=4)
    vpo_ = R.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
    vpo_["w"] = vp["w"]
    theta = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
    opts = vbmc_amd.optimize.DEFAULT_OPTIONS
    NSent = int(np.ceil(opts["NSent"](K) / K)) if callable(opts["NSent"]) else int(np.ceil(opts["NSent"] / K))
    NSfine = int(np.ceil(opts["NSentFine"](K) / K)) if callable(opts["NSentFine"]) else int(np.ceil(opts["NSentFine"] / K))
    rng = np.random.default_rng(0)

    def timeit(f, n):
        f()
        t1 = time.perf_counter()
        for _ in range(n):
            f()
        return (time.perf_counter() - t1) / n

    t_adam = timeit(lambda: R.negelcbo_vbmc(theta, 0, vpo_, gpo, NSent, True, 0, rng=rng), 3)
    t_sieve = timeit(lambda: R.negelcbo_vbmc(theta, 0, vpo_, gpo, 0, False, 0), 3)
    t_full = timeit(lambda: R.negelcbo_vbmc(theta, 0, vpo_, gpo, NSfine, False, 1, separate_K=True, rng=rng), 1)
    P("")
    P("NumPy oracle (line-by-line restatement, 1 process on the box's host) per evaluation of the same kinds: Adam-loop call "
      "(`Ns = %d` per component, value + gradient) **%.3f s**, sieve call (deterministic entropy, value) %.3f s, full-variance ELCBO "
      "(`Ns = %d` per component, `separate_K`) %.2f s.\n" % (NSent, t_adam, t_sieve, NSfine, t_full))
    P("| Nfastopts | device wall s | the same evaluations at the oracle's cost, s | ratio |\n|---|---|---|---|")
    for nfast, dt, n_sieve, n_adam, n_full, _, _ in rows:
        host = n_sieve * t_sieve + n_adam * t_adam + n_full * t_full
        P("| %d | %.3f | %.0f | %.0f x |" % (nfast, dt, host, host / dt))
    P("")
    open(out, "w").write("\n".join(L))
    print("\n".join(L))


if __name__ == "__main__":
    main()
