#!/bin/bash
# Round profile on the MI355X box: full GPU test suite, bench line, rocprofv3 kernel trace and two PMC passes of the
# same bench command, aux benches.  Writes gpurun_out/<tag>/ (summaries only; the rocpd databases are deleted).
#   usage (through gpurun):  bash tools/profile_round.sh p4
set -u
tag=${1:-px}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -4 > $out/pytest_gpu.txt   # (RCCL prints a banner after pytest's last line)
python bench.py --extras 2> $out/bench_stderr.txt | tail -1 > $out/bench.json
python tools/bench_aux.py 2>/dev/null | tail -1 > $out/bench_aux.json
vbmc_amd/lib/microbench > $out/microbench.json 2>&1
[ -f profiles/isa_meta_qs3.txt ] && cp profiles/isa_meta_qs3.txt $out/isa_meta.txt
rocprofv3 --kernel-trace --stats -d $out/t -o p -- python bench.py --no-cpu-baseline --no-aux > $out/bench_traced.json 2> $out/trace_stderr.txt
python tools/rocpd_summary.py $(find $out/t -name '*.db' | head -1) > $out/kernel_trace.md
rm -rf $out/t
rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $out/a -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-aux > /dev/null 2> $out/pmc_a_stderr.txt
python tools/pmc_summary.py $(find $out/a -name '*.db' | head -1) > $out/pmc_a.md
rm -rf $out/a
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F64 -d $out/b -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-aux > /dev/null 2> $out/pmc_b_stderr.txt
python tools/pmc_summary.py $(find $out/b -name '*.db' | head -1) > $out/pmc_b.md
rm -rf $out/b
tail -2 $out/pytest_gpu.txt; cat $out/bench.json | cut -c1-400; head -8 $out/kernel_trace.md
