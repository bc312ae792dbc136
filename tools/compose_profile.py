#!/usr/bin/env python
"""Assemble profiles/<round>_<tag>_summary.md (and refresh profiles/<round>_pmc.json) from the files tools/profile_round.sh left
in gpurun_out/<tag>/.   usage: [ROUND=r02] python tools/compose_profile.py p1 "what changed since the previous profile" """
import hashlib
import json
import os
import re
import subprocess
import sys

RND = os.environ.get("ROUND", "r02")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402  (code-only hash: comments and white space do not count)

tag, changes = sys.argv[1], sys.argv[2]
o = "gpurun_out/%s/" % tag
rd = lambda f: open(o + f).read().strip()  # noqa: E731


def last_json(txt):
    """the JSON line of a bench run's stdout (RCCL prints a banner after it when the process exits)"""
    return [ln for ln in txt.splitlines() if ln.startswith("{")][-1]

pa, pb = rd("pmc_a.md"), rd("pmc_b.md")


def grab(txt, kern, ctr):
    seg = txt[txt.index(kern):][:1500]
    return float(re.search(r"- %s = ([0-9.e+]+)" % ctr, seg).group(1))


kern = "`void k_entropy_mfma<3, 3, true, false, 1, 1, false, false, false>"   # the headline instantiation: QS 3, three k-tiles + component tail, device RNG, chunk grid
kern_walk = "`void k_entropy_mfma<3, 3, true, false, 1, 1, false, false, true>"   # ... and its walking launch (round 6: blocking calls)
fetch, write = grab(pa, kern, "FETCH_SIZE"), grab(pb, kern, "WRITE_SIZE")
hbm = int(round(fetch * 1024 * 2 + write * 1024))
walk = {}
if kern_walk in pa and kern_walk in pb:
    wf, ww = grab(pa, kern_walk, "FETCH_SIZE"), grab(pb, kern_walk, "WRITE_SIZE")
    walk = {"walk_FETCH_SIZE_KB_per_launch": wf, "walk_WRITE_SIZE_KB_per_launch": ww, "walk_hbm_bytes_per_launch": int(round(wf * 1024 * 2 + ww * 1024))}
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
json.dump({"profile": "profiles/%s_%s_summary.md" % (RND, tag), "kernel": "k_entropy_mfma<3,3,true,false,1,1>", "commit": commit,
           "kernel_source_sha256_16": kernel_source_hash(),
           "workload": "python bench.py (R=64, C3, device RNG)", "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
           "correction": "FETCH_SIZE doubled (gfx950 counts 128-B read requests at 64 B, MI355X_MICROARCH.md HBM section); "
                         "WRITE_SIZE as reported (uncalibrated)",
           "hbm_bytes_per_launch": hbm, **walk}, open("profiles/%s_pmc.json" % RND, "w"), indent=1)
busy, act = grab(pa, kern, "SQ_VALU_MFMA_BUSY_CYCLES"), grab(pa, kern, "SQ_ACTIVE_INST_VALU")
n_mfma, n_valu = grab(pa, kern, "SQ_INSTS_MFMA"), grab(pa, kern, "SQ_INSTS_VALU")
TS = 64 * 50 * 313 * 2.0     # tile-signs per launch at the headline shape: R x K x ceil(5000 / 16) tiles x 2 signs
b = json.loads(last_json(rd("bench.json")))
b.update(b.get("aux", {}))   # round 2: the auxiliary legs are nested under "aux"
bt = json.loads(last_json(rd("bench_traced.json")))
aux = json.loads(last_json(rd("bench_aux.json")))
r = b["roofline"]
mf, va = busy / 1024 / 2.4e9 * 1e3, act * 4 / 1024 / 2.4e9 * 1e3
isa = rd("isa_meta.txt") if os.path.exists(o + "isa_meta.txt") else "(not collected)"
pc = rd("pmc_c.md") if os.path.exists(o + "pmc_c.md") else None
lane_txt, cfg_txt = "", ""
if pc:
    thr, actc = grab(pc, kern, "SQ_THREAD_CYCLES_VALU"), grab(pc, kern, "SQ_ACTIVE_INST_VALU")
    coex = grab(pc, kern, "SQ_VALU_MFMA_COEXEC_CYCLES")
    lane_txt = f"""
## PMC per launch, pass C (round 4: VALU lane utilisation, instruction classes)

{pc}

VALU lane utilisation of the headline kernel = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = {thr:.4g} / (64 x {actc:.4g}) =
**{thr / (64 * actc):.3f}** (1.0 = every VALU instruction runs with all 64 lanes enabled); MFMA / VALU co-execution
(`SQ_VALU_MFMA_COEXEC_CYCLES`) {coex:.4g} cycles against {busy:.4g} MFMA-busy cycles.
"""
for name, label in (("c1", "BASELINE configs[1]: D=6 N=200 K=10 Ns=1000 S=8, R=64"), ("c4", "BASELINE configs[4]: D=20 N=800 K=100 Ns=20000 S=20, R=16")):
    if os.path.exists(o + "kernel_trace_%s.md" % name):
        bl = last_json(rd("bench_%s.json" % name))
        cfg_txt += f"""
## {label}: `rocprofv3 --kernel-trace --stats` and one `--pmc` pass of `python bench.py <shape flags> --no-cpu-baseline --no-aux`

```
{bl[:1500]}
```

{rd('kernel_trace_%s.md' % name)}

{rd('pmc_%s.md' % name)}
"""
txt = f"""# Round {int(RND[1:])}, profile {tag[1:]}

Produced by `bash tools/profile_round.sh {tag}` on the MI355X box (`cd /tmp && export TMPDIR=/tmp` first): full GPU
test suite, `python bench.py --extras`, `tools/bench_aux.py`, `tools/microbench.hip`, then
`rocprofv3 --kernel-trace --stats` and two separate `--pmc` passes of
`python bench.py [--steps 3 --warmup 1] --no-cpu-baseline --no-aux` (the timed region of the default command; the auxiliary legs and the CPU baseline run after it); assembled by `tools/compose_profile.py`.  {changes}
In a blocking call `k_logjoint_mfma` runs on the context's second, lower-priority stream beside the entropy kernel: its traced duration there
is the span over which its workgroups were fitted into the entropy kernel's idle slots (alone it takes 0.12 ms); in the pipelined steps it runs
at the head of its pass on the slot stream, beside the other pass's entropy kernel.

`pytest tests -m gpu`: **{rd('pytest_gpu.txt').splitlines()[-1]}**.

## Bench line (`python bench.py --extras`, un-profiled)

```
{rd('bench.json')}
```

## Bench line of the kernel-trace run

```
{last_json(rd('bench_traced.json'))}
```

## Kernel trace (tools/rocpd_summary.py; 92 ELBO launches = 11 warm-up + 20 timed (chunk grid) + 21 of the --sync-steps leg (walking launches) + 30 roofline-leg (chunk grid) + 10 walking launches alone; k_chol / k_gp_* / k_alpha_solve = the one-off gplite_post that builds the synthetic GP posterior, outside the timed region)

{rd('kernel_trace.md')}

### The dominant kernel's launches by phase of the command (tools/rocpd_launches.py)

In the pipelined steps four batches are in flight on two streams: the entropy kernel of one batch shares the chip with the log joint and
the small kernels of the next, so its span is longer than its work and the spans of consecutive launches overlap (their sum exceeds the wall
time).  `roofline.kernel_ms` of the bench line is the last phase -- the kernel alone on the device, `vbmc_ctx_set_profiling(ctx, 2)` -- and
that is the row it has to agree with.

{rd('kernel_phases.md') if os.path.exists(o + 'kernel_phases.md') else '(not collected in this run)'}

## Registers / LDS / scratch of the entropy kernels from the compiler's own metadata (`tools/isa_meta.py 3`)

The `vgpr_count` column of the rocprofv3 table above is the dispatch packet's arch-VGPR allocation field, not the register
count (VERDICT r1 asked: 128 there vs 255 claimed); the authoritative figures are the `.amdgpu_metadata` of the code object:

```
{isa}
```

## PMC per launch (R = 64 evaluations), pass A

{pa}

## PMC per launch, pass B

{pb}

{lane_txt}{cfg_txt}
## Other device paths at the C3 GP shape (`tools/bench_aux.py`, wall time per call incl. H2D/D2H)

```
{rd('bench_aux.json')}
```

## Microbenchmarks (`tools/microbench.hip`)

```
{rd('microbench.json')}
```

## Reading

* Step = {b['ms_per_step']:.2f} ms for R = 64 evaluations -> **{b['value'] / 1e3:.1f} k ELBO+grad evals/s** on one MI355X
  ({bt['value'] / 1e3:.1f} k in the traced run); C port of the MATLAB loop nest on the box's host: {b['cpu_baseline']['value']:.2f} evals/s on 1 core
  ({b['cpu_baseline']['all_cores']['value']:.1f} with OpenMP on {b['cpu_baseline']['all_cores']['cores']} threads).
* `k_entropy_mfma` {r['kernel_ms']:.2f} ms by HIP events in `bench.py` (rocprofv3 average in the table above): per tile-sign (800
  sample x component pairs) {n_mfma / TS:.1f} MFMA (6 S-step, the antithetic pair sharing the even part, + 13 PV: three k-tiles and the
  two-component tail) + {n_valu / TS:.0f} VALU instructions (`SQ_INSTS_MFMA`, `SQ_INSTS_VALU` over {TS:.4g} tile-signs).  MFMA busy {busy:.4g} cycles / 1024 SIMDs = {mf:.2f} ms,
  VALU active {act:.3g} x 4 / 1024 = {va:.2f} ms; the two do not overlap for fp64 -> fp64 pipe ~{100 * (mf + va) / r['kernel_ms']:.0f} % busy.
  Register-limited to 2 waves/SIMD.
* Algorithmic flops (SURVEY 8d) 9.29e10 per launch -> {r['achieved']:.1f} TFLOP/s = **{100 * r['frac']:.0f} % of the 78.6 TFLOP/s dense fp64 peak**
  (exponentials -- 1.6e9 per launch, 8 fp64 + 3 int ops each -- and tile padding not counted).
* HBM: FETCH_SIZE {fetch:.0f} KB (x2, gfx950 correction) + WRITE_SIZE {write:.0f} KB = {hbm / 1e6:.1f} MB per launch
  (`profiles/{RND}_pmc.json`), < 0.1 % of HBM bandwidth: per-chunk partial records and the packed mixture parameters.
* On-device Adam, one chain: {b['device_adam_R1_evals_per_s'] / 1e3:.1f} k evals/s ({b['host_loop_R1_evals_per_s'] / 1e3:.1f} k with one host call per evaluation
  through the prepared objective); two chains in lock-step {b['device_adam_R2_evals_per_s'] / 1e3:.1f} k; parity mode (fp64 eps streamed
  from HBM, {b['eps_streamed']['eps_stream_GBps']:.0f} GB/s) {b['eps_streamed']['evals_per_s'] / 1e3:.1f} k evals/s; block-sparse mode
  {b['block_sparse']['evals_per_s'] / 1e3:.1f} k evals/s with identical output.
* GP side (`tools/bench_aux.py`): `gplite_post` (S = 20, N = 400) {aux['gplite_post_ms']:.1f} ms; `gplite_pred` 8192 x 20: {aux['gplite_pred_8192_ms']:.2f} ms wall;
  acquisition sweep on 8192 points: `acqf` {aux['acqwrapper_acqf_8192_ms']:.2f} ms, VIQR with 100 importance points {aux['acqwrapper_acqviqr_8192_Na100_ms']:.2f} ms;
  `eval_fullelcbo` {aux['eval_fullelcbo_ms']:.2f} ms; value-only `entlb` sieve of 250 candidates {aux['entlb_sieve_R250_ms']:.2f} ms;
  `gplite_nlZ`+gradient {aux.get('nlz_grad_B1_evals_per_s', 0):.0f} / {aux.get('nlz_grad_B16_evals_per_s', 0) / 1e3:.1f} k / {aux.get('nlz_grad_B64_evals_per_s', 0) / 1e3:.1f} k / {aux.get('nlz_grad_B256_evals_per_s', 0) / 1e3:.1f} k evals/s at B = 1 / 16 / 64 / 256.
"""
open("profiles/%s_%s_summary.md" % (RND, tag), "w").write(txt)
print("wrote profiles/%s_%s_summary.md" % (RND, tag), len(txt))
