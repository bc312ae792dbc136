#!/bin/bash
# Round-6 profile of the paths that had no kernel trace of their own: the deterministic-entropy sieve (R = 250),
# eval_fullelcbo, the IMIQR importance sampler, gplite_pred (trace + FETCH_SIZE / WRITE_SIZE passes) and the acquisition sweeps on it.
#   usage (through gpurun):  bash tools/profile_aux.sh r6aux
set -u
tag=${1:-aux}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for leg in sieve fullelcbo pred acqf; do
  python tools/prof_aux_leg.py $leg 2>/dev/null | tail -1 > $out/wall_$leg.txt
  rocprofv3 --kernel-trace --stats -d $out/t_$leg -o p -- python tools/prof_aux_leg.py $leg > /dev/null 2> $out/trace_${leg}_stderr.txt
  python tools/rocpd_summary.py $(find $out/t_$leg -name '*.db' | head -1) > $out/kernel_trace_$leg.md
  python tools/rocpd_timeline.py $(find $out/t_$leg -name '*.db' | head -1) 40 > $out/timeline_$leg.md 2>/dev/null
  rm -rf $out/t_$leg
done
python tools/prof_imiqr_sampler.py 2>/dev/null | tail -3 > $out/wall_imiqr.txt
rocprofv3 --kernel-trace --stats -d $out/t_imiqr -o p -- python tools/prof_imiqr_sampler.py > /dev/null 2> $out/trace_imiqr_stderr.txt
python tools/rocpd_summary.py $(find $out/t_imiqr -name '*.db' | head -1) > $out/kernel_trace_imiqr.md
rm -rf $out/t_imiqr
# gplite_pred: HBM traffic of its kernels (separate passes, as the guide prescribes)
rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $out/pa -o p -- python tools/prof_aux_leg.py pred > /dev/null 2> $out/pmc_pred_a_stderr.txt
python tools/pmc_summary.py $(find $out/pa -name '*.db' | head -1) k_pred > $out/pmc_pred_a.md
rm -rf $out/pa
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $out/pb -o p -- python tools/prof_aux_leg.py pred > /dev/null 2> $out/pmc_pred_b_stderr.txt
python tools/pmc_summary.py $(find $out/pb -name '*.db' | head -1) k_pred > $out/pmc_pred_b.md
rm -rf $out/pb
# the 3.8 ms readings of gplite_pred_8192_ms: call by call, fresh engine, results into new arrays
python tools/pred_calls.py > $out/pred_calls.txt 2>/dev/null
cat $out/wall_*.txt; head -12 $out/kernel_trace_sieve.md; cat $out/pred_calls.txt
