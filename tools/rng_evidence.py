"""Statistical evidence for the device normal stream (Philox4x32-7 + Box-Muller on 24-bit uniforms, device_math.h) that stands
in for MATLAB's randn in entmc_vbmc (ent/entmc_vbmc.m:53).  VERDICT r1 item 5.  Run on the GPU box:

    python tools/rng_evidence.py [out.md]        (default: gpurun_out/rng_evidence.md; copy to profiles/)

1. Goodness of fit of >= 1e7 draws: Kolmogorov-Smirnov and Anderson-Darling against N(0,1), moments.
2. Tail mass beyond 3, 4, 5 sigma against the exact normal tail with a binomial interval (the 24-bit radius uniform caps |z|
   at sqrt(-2 ln 2^-25) = 5.89: expected loss of mass beyond that is 3.9e-9 per draw).
3. Independence across every counter of the stream -- dimension within a Philox block, dim-block, sample, component,
   restart, seed -- as Pearson correlations of the values and of their squares against 1/sqrt(n).
4. What matters downstream: the distribution of the entropy estimate H (and of the ELBO gradient norm) over 200 seeds at
   the headline shape with device draws against the same kernel fed NumPy's fp64 normals: Welch test on the means, Levene
   test on the variances.
"""
import os
import sys
import time

import numpy as np
from scipy import stats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "rng_evidence.md")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    eng = vbmc_amd.Engine(0)
    L = []
    P = L.append
    P("# Device normal stream: statistical evidence (tools/rng_evidence.py)\n")
    P("Stream: Philox4x32-7 keyed by the seed, counter (sample, component, restart, dim-block), four normals per block by "
      "Box-Muller on 24-bit uniforms with the hardware log/sqrt/sin/cos (`vbmc_amd/csrc/device_math.h`), dumped by `vbmc_rng_dump` "
      "(the same device function the entropy kernel calls).\n")
    fails = []

    # ---- 1. goodness of fit
    D, K, R, Ns = 10, 50, 4, 10000
    t0 = time.time()
    x = np.concatenate([eng.ctx.rng_dump(D, K, R, Ns, seed).reshape(-1) for seed in (1, 2)])
    n = x.size
    ks = stats.kstest(x, "norm")
    ad = stats.anderson(x[: 5_000_000], "norm")     # scipy's AD uses estimated mean / sd: critical values for that case
    m1, m2, m3, m4 = np.mean(x), np.var(x), stats.skew(x), stats.kurtosis(x)
    P("## 1. Goodness of fit, n = %d draws (D=%d, K=%d, R=%d, Ns=%d, seeds 1 and 2)\n" % (n, D, K, R, Ns))
    P("| statistic | value | reference |\n|---|---|---|")
    P("| Kolmogorov-Smirnov D | %.3e | p = %.3f (reject below 0.01); 1 %% critical value 1.63/sqrt(n) = %.3e |" % (ks.statistic, ks.pvalue, 1.63 / np.sqrt(n)))
    P("| Anderson-Darling A^2 (first 5e6 draws, estimated mean/sd) | %.3f | critical values %s at %s %% |"
      % (ad.statistic, np.array2string(ad.critical_values, precision=3), np.array2string(ad.significance_level)))
    P("| mean | %+.2e | 0 +- %.1e (2 s.e.) |" % (m1, 2 / np.sqrt(n)))
    P("| variance | %.6f | 1 +- %.1e |" % (m2, 2 * np.sqrt(2.0 / n)))
    P("| skewness | %+.2e | 0 +- %.1e |" % (m3, 2 * np.sqrt(6.0 / n)))
    P("| excess kurtosis | %+.2e | 0 +- %.1e |" % (m4, 2 * np.sqrt(24.0 / n)))
    P("| distinct values / n | %.4f | a 24 x 24-bit grid: ties are possible but rare |\n" % (np.unique(x).size / n))
    if ks.pvalue < 0.01:
        fails.append("KS p = %.4f" % ks.pvalue)
    if ad.statistic > ad.critical_values[-1]:
        fails.append("AD %.3f > %.3f" % (ad.statistic, ad.critical_values[-1]))
    for nm, v, tol in (("mean", m1, 4 / np.sqrt(n)), ("variance", m2 - 1, 4 * np.sqrt(2.0 / n)), ("skew", m3, 4 * np.sqrt(6.0 / n)),
                       ("kurtosis", m4, 4 * np.sqrt(24.0 / n))):
        if abs(v) > tol:
            fails.append("%s off by %.2e (> 4 s.e. %.2e)" % (nm, v, tol))

    # ---- 2. tails
    P("## 2. Tail mass (two-sided), n = %d\n" % n)
    P("| threshold | observed count | expected count | z-score |\n|---|---|---|---|")
    for thr in (2.0, 3.0, 4.0, 4.5, 5.0):
        p = 2 * stats.norm.sf(thr)
        obs = int(np.sum(np.abs(x) > thr))
        z = (obs - n * p) / np.sqrt(n * p * (1 - p))
        P("| %.1f sigma | %d | %.1f | %+.2f |" % (thr, obs, n * p, z))
        if abs(z) > 4:
            fails.append("tail beyond %.1f sigma: z = %.2f" % (thr, z))
    P("| max abs z | %.4f | cap sqrt(-2 ln 2^-25) = 5.887 | |\n" % np.max(np.abs(x)))

    # ---- 3. independence across the counters
    P("## 3. Correlations across the stream's counters (values / squares; |r| should be ~ 1/sqrt(n))\n")
    P("| pair | n pairs | r(values) | r(squares) | 3/sqrt(n) |\n|---|---|---|---|---|")
    A = eng.ctx.rng_dump(12, 8, 6, 80000, 7)   # (R, K, Mh, D): D = 12 -> three full dim-blocks of four

    def corr(a, b, name):
        a, b = a.reshape(-1), b.reshape(-1)
        r1 = np.corrcoef(a, b)[0, 1]
        r2 = np.corrcoef(a * a, b * b)[0, 1]
        lim = 3 / np.sqrt(a.size)
        P("| %s | %d | %+.2e | %+.2e | %.1e |" % (name, a.size, r1, r2, lim))
        if abs(r1) > 1.5 * lim or abs(r2) > 1.5 * lim:
            fails.append("correlation %s: %.2e / %.2e (limit %.1e)" % (name, r1, r2, 1.5 * lim))

    corr(A[..., 0], A[..., 1], "dims 0,1 of one Philox block (the cos / sin pair of one Box-Muller)")
    corr(A[..., 0], A[..., 2], "dims 0,2 of one block (two Box-Muller pairs)")
    corr(A[..., 1], A[..., 3], "dims 1,3 of one block")
    corr(A[..., 0:4], A[..., 4:8], "dim-block q and q+1")
    corr(A[:, :, :-1, :], A[:, :, 1:, :], "sample i and i+1")
    corr(A[:, :, :-16, :], A[:, :, 16:, :], "sample i and i+16 (next tile)")
    corr(A[:, :-1], A[:, 1:], "component j and j+1")
    corr(A[:-1], A[1:], "restart r and r+1")
    B = eng.ctx.rng_dump(12, 8, 6, 80000, 8)
    corr(A, B, "seed s and s+1 (consecutive Adam iterations)")
    Bh = eng.ctx.rng_dump(12, 8, 6, 80000, 7 + (1 << 32))
    corr(A, Bh, "seeds differing in the high key word only")
    P("")

    # ---- 4. the entropy estimate over seeds: device draws vs NumPy fp64 normals through the same kernel
    D, N, K, S, Ns = 10, 400, 50, 20, 10000
    inp = synth_inputs(0, D, N, K, S)
    gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng)
    vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
    vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
    theta = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
    nseed = 200
    Hd, Hn, Gd, Gn = [], [], [], []
    rng = np.random.default_rng(12345)
    for s in range(nseed):
        r = vbmc_amd.negelcbo_batch(theta, 0, vp, gp, Ns, True, 0, seed=1000 + s, engine=eng, outputs=("H", "dH"))
        Hd.append(r["H"][0]); Gd.append(np.linalg.norm(r["dH"][:, 0]))
        eps = rng.standard_normal((K, Ns // 2, D))
        r = vbmc_amd.negelcbo_batch(theta, 0, vp, gp, Ns, True, 0, eps=eps, engine=eng, outputs=("H", "dH"))
        Hn.append(r["H"][0]); Gn.append(np.linalg.norm(r["dH"][:, 0]))
    Hd, Hn, Gd, Gn = map(np.asarray, (Hd, Hn, Gd, Gn))
    P("## 4. Entropy estimate H and |dH| over %d seeds at the headline shape (D=%d, K=%d, Ns=%d per component)\n" % (nseed, D, K, Ns))
    P("| quantity | device stream: mean +- sd | NumPy fp64 normals: mean +- sd | Welch p (means) | Levene p (variances) |\n|---|---|---|---|---|")
    for nm, a, b in (("H", Hd, Hn), ("norm(dH)", Gd, Gn)):
        pw = stats.ttest_ind(a, b, equal_var=False).pvalue
        pl = stats.levene(a, b).pvalue
        P("| %s | %.6f +- %.2e | %.6f +- %.2e | %.3f | %.3f |" % (nm, a.mean(), a.std(ddof=1), b.mean(), b.std(ddof=1), pw, pl))
        if pw < 0.002 or pl < 0.002:
            fails.append("%s over seeds: Welch p %.4f, Levene p %.4f" % (nm, pw, pl))
    P("")
    P("Result: **%s**%s  (%.0f s)\n" % ("PASS" if not fails else "FAIL", "" if not fails else " -- " + "; ".join(fails), time.time() - t0))
    open(out, "w").write("\n".join(L))
    print("\n".join(L))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
