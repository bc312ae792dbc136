#!/bin/bash
# Round profile on the MI355X box: full GPU test suite, bench line, rocprofv3 kernel trace and two PMC passes of the
# same bench command, aux benches.  Writes gpurun_out/<tag>/ (summaries only; the rocpd databases are deleted).
#   usage (through gpurun):  bash tools/profile_round.sh p4
set -u
tag=${1:-px}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "${ONLY_TRACE:-0}" = "1" ]; then
  rocprofv3 --kernel-trace --stats -d $out/t -o p -- python bench.py --no-cpu-baseline --no-aux > $out/bench_traced.json 2> $out/trace_stderr.txt
  python tools/rocpd_summary.py $(find $out/t -name '*.db' | head -1) > $out/kernel_trace.md
  python tools/rocpd_launches.py $(find $out/t -name '*.db' | head -1) k_entropy_mfma "pipelined steps (8 + 3 warm-up, 20 timed):31" "blocking calls, log joint forked beside it (21 of the --sync-steps leg, walking launches; 10 of the roofline leg, chunk grid):31" "roofline leg, kernel alone, chunk grid as in the pipelined steps (the figure bench.py prices):20" "the blocking call's walking launch, kernel alone:10" > $out/kernel_phases.md
  rm -rf $out/t
  cat $out/kernel_phases.md
  exit 0
fi
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -4 > $out/pytest_gpu.txt   # (RCCL prints a banner after pytest's last line)
python bench.py --extras 2> $out/bench_stderr.txt | grep '^{' | tail -1 > $out/bench.json   # (RCCL prints a banner after the line when the process exits)
python tools/bench_aux.py 2>/dev/null | grep '^{' | tail -1 > $out/bench_aux.json
vbmc_amd/lib/microbench > $out/microbench.json 2>&1
[ -f profiles/isa_meta_qs3.txt ] && cp profiles/isa_meta_qs3.txt $out/isa_meta.txt
rocprofv3 --kernel-trace --stats -d $out/t -o p -- python bench.py --no-cpu-baseline --no-aux > $out/bench_traced.json 2> $out/trace_stderr.txt
python tools/rocpd_summary.py $(find $out/t -name '*.db' | head -1) > $out/kernel_trace.md
# the dominant kernel's launches by phase of the command: 8 + 3 warm-up and 20 timed steps (pipelined: four batches in flight on two streams, the
# kernels of consecutive batches overlap), then 21 + 10 blocking calls (log joint forked beside the kernel) and 20 with the kernel alone
python tools/rocpd_launches.py $(find $out/t -name '*.db' | head -1) k_entropy_mfma "pipelined steps (8 + 3 warm-up, 20 timed):31" "blocking calls, log joint forked beside it (21 of the --sync-steps leg, walking launches; 10 of the roofline leg, chunk grid):31" "roofline leg, kernel alone, chunk grid as in the pipelined steps (the figure bench.py prices):20" "the blocking call's walking launch, kernel alone:10" > $out/kernel_phases.md
rm -rf $out/t
rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $out/a -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-aux > /dev/null 2> $out/pmc_a_stderr.txt
python tools/pmc_summary.py $(find $out/a -name '*.db' | head -1) > $out/pmc_a.md
rm -rf $out/a
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F64 -d $out/b -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-aux > /dev/null 2> $out/pmc_b_stderr.txt
python tools/pmc_summary.py $(find $out/b -name '*.db' | head -1) > $out/pmc_b.md
rm -rf $out/b
# round 4: VALU lane utilisation and the instruction classes of the headline kernel (third PMC pass), and a kernel trace + one PMC pass for
# the two other single-GPU configurations of BASELINE.json (configs[1] and configs[4]) at the bench's own command lines
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64 -d $out/c -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-aux > /dev/null 2> $out/pmc_c_stderr.txt
python tools/pmc_summary.py $(find $out/c -name '*.db' | head -1) > $out/pmc_c.md
rm -rf $out/c
for cfg in "c1 --D 6 --N 200 --K 10 --Ns 1000 --S 8" "c4 --D 20 --N 800 --K 100 --Ns 20000 --restarts 16 --steps 6"; do
  set -- $cfg; name=$1; shift
  rocprofv3 --kernel-trace --stats -d $out/t$name -o p -- python bench.py "$@" --no-cpu-baseline --no-aux > $out/bench_$name.json 2> $out/trace_${name}_stderr.txt
  python tools/rocpd_summary.py $(find $out/t$name -name '*.db' | head -1) > $out/kernel_trace_$name.md
  rm -rf $out/t$name
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $out/p$name -o p -- python bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-aux > /dev/null 2> $out/pmc_${name}_stderr.txt
  python tools/pmc_summary.py $(find $out/p$name -name '*.db' | head -1) k_entropy > $out/pmc_$name.md
  rm -rf $out/p$name
done
tail -2 $out/pytest_gpu.txt; cat $out/bench.json | cut -c1-400; head -8 $out/kernel_trace.md
