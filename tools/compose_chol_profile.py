#!/usr/bin/env python
"""Assemble profiles/r02_chol.md from gpurun_out/chol/ (written on the MI355X box by the command quoted at the top of that file)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
o = os.path.join(ROOT, "gpurun_out", "chol")
cb = open(os.path.join(o, "chol_bench.txt")).read()
lines = [ln for ln in cb.split("\n") if not ln.startswith("  g")]
rows = [json.loads(ln) for ln in lines if ln.startswith("{")]
tab = "| N | matrices | ms (median) | max rel. err vs host long-double Cholesky | failure index on an indefinite copy (got, expected) |\n|---|---|---|---|---|\n"
for r in rows:
    tab += "| %d | %d | %.4f | %.1e | %d, %d |\n" % (r["N"], r["S"], r["ms_median"], r["max_rel_err"], r["p_indefinite"][0], r["p_indefinite"][1])
stamps = "\n".join(ln for ln in lines if re.match(r"^(step:|\s+\d+:|sum panel|look-ahead|shader clock|tile stream)", ln))
trace = open(os.path.join(o, "nlz_trace.md")).read().split("\n\n")[0]
gp = "\n".join(ln for ln in open(os.path.join(o, "gp_post.txt")).read().split("\n") if ln.startswith("gplite_post"))
aux = json.loads(open(os.path.join(o, "bench_aux.json")).read().strip().split("\n")[-1])
txt = f"""# Round 2: the Cholesky kernel, second generation (`k_chol2`, `vbmc_amd/csrc/chol_mfma.h`) and the single-vector solve

`k_chol` (0.52 ms at N = 400) was the floor of every GP entry point: 55 % of a single `gplite_nlZ` evaluation (the slice-sampling
chain of `gplite_train` is a SEQUENCE of such evaluations) and 65 % of the device time of `gplite_post`.  Everything below is from
`tools/chol_bench.hip` (stand-alone harness: random SPD matrix, result against a host long-double Cholesky, MATLAB's failure index on
an indefinite copy, HIP-event times, and with `-DCHOL_TS` the phase stamps inside the kernel) on the MI355X box; collected with
`bash tools/profile_chol.sh` (chol_bench over sizes, its phase stamps and tile-stream mode, `tools/bench_aux.py`, `tools/gp_post_probe.py`,
`rocprofv3 --kernel-trace --stats -- python tools/prof_nlz.py` summarised by `tools/rocpd_summary.py`)
and assembled by `tools/compose_chol_profile.py`.

## What changed, with the measurement that motivated each step (N = 400, one matrix)

| step | ms | what the stamps showed |
|---|---|---|
| first generation (`k_chol`, round 1) | 0.524 | panel 152 us (16-step substitution per column), update 347 us; the panel operands were fetched through ONE flat pointer (LDS or global scratch): `flat_load` + `s_waitcnt vmcnt(0) lgkmcnt(0)` in front of every MFMA pair, 18 spilled VGPRs |
| panel as an MFMA product with inv(Rkk), `ds_read` operands, LDS work counter | 0.399 | panel 66 us; update still 305 us |
| fast path without masks, straight-line groups, rsq-based pivots | 0.344 | update 248 us |
| static deal (no LDS atomic at the head of a group), wave-uniform tile walk in SGPRs, 32-byte panel vectors, SGPR-base addressing | 0.309 | update 221 us: per group of 4 tiles 1000 cycles to ISSUE 16 loads, ~2200 waiting for them -- the memory pipe of the CU is saturated |
| `chol_bench 9`: one CU streams such tiles at 75 GB/s (loads) / 65 GB/s (load + store) | | the update of the first steps IS at that bound; software pipelining, more waves, register double buffering: no change (all measured) |
| diagonal tile factored in the accumulator layout, 4 pivots + one rank-4 MFMA at a time; shorter pivot chain | 0.296 | look-ahead 7.3 -> 5.4 us per step (factor 10 700 -> 6 200 cycles) |
| two panels per pass over the trailing matrix (half the traffic), wave 4 idle when the look-ahead is critical | **0.283** | update 198 us; 256 matrices at once: 0.66 -> **0.35 ms** (that case was bound by L2/HBM traffic: 3 GB per launch) |
| one step of iterative refinement in the panel (P += W (A - Rkk' P), 8 more MFMAs per panel tile) | **0.291** | accuracy, not speed: the product with the explicit inverse is accurate to cond(Rkk) eps only -- the 15x random-shape sweep found alpha off by 2e-8 on a kernel matrix of condition 1e7 (tolerance 1e-8; the substitution it replaced is backward stable); with the refinement the 40x sweep and `test_posterior_on_ill_conditioned_kernel_matrices` pass; 256 matrices: 0.38 ms |

`k_alpha_solve1` (alpha = R \\ (R' \\ (y - m)), one right-hand side): 92 -> 55 us at N = 400.  The old kernel spent a 64-lane reduction,
two barriers and two L2 round trips per block step; the new one keeps the vector element of a thread in a register, fetches the
next step's 16 matrix entries across an LDS-only barrier (`s_waitcnt lgkmcnt(0); s_barrier` -- `__syncthreads()` also drains the
loads), and reads the forward pass's columns with 16-byte loads (a lane per column touches 64 cache lines per instruction: the
texture addresser, not the latency, set the pace; 72 -> 55 us from that alone).

## Final Cholesky kernel, all sizes (`chol_bench 1 N S`)

{tab}
N <= 592: two panels in LDS; N <= 1200: one panel; beyond: panel in a global scratch block (one workgroup per matrix stays the design:
VBMC's training sets are a few hundred points).

## Phase stamps of the final kernel, N = 400, one matrix (us; even steps = panel + look-ahead + next row block, odd steps = rank-32 update)

```
{stamps}
```

## What it buys (`tools/bench_aux.py`, `tools/gp_post_probe.py`, `tools/prof_nlz.py` under rocprofv3)

* `gplite_nlZ` for ONE hyper-parameter vector: value {aux['nlz_value_B1_evals_per_s']:.0f} evals/s (start of the round: 1 330), value + gradient {aux['nlz_grad_B1_evals_per_s']:.0f} (1 040);
  256 vectors per call: value {aux['nlz_value_B256_evals_per_s']/1e3:.0f} k evals/s (236 k), value + gradient {aux['nlz_grad_B256_evals_per_s']/1e3:.0f} k (102 k).
* `gplite_post` (S = 20, N = 400), steady state of 20 calls (1.06 ms with the factors left on the device and 1.52 ms with the 25.6 MB
  read back before this work; besides the kernels: the surrogate's uploads are queued and awaited once, the factors move in one D2D copy):
```
{gp}
```
* Kernel trace of 10 `gplite_nlZ` + gradient calls each at B = 1, 64, 256:

{trace}

  `k_tri_inverse` + `k_syrk_tt` (inv(K) = T'T for the gradient) are now the larger part of a batched gradient evaluation.
"""
open(os.path.join(ROOT, "profiles", "r02_chol.md"), "w").write(txt)
print("wrote profiles/r02_chol.md", len(txt))
