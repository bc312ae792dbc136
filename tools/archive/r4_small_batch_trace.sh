#!/bin/bash
# Kernel timeline of the pipelined step at a small batch (R restarts on one device: the strong-scaling share of BASELINE configs[3]).
#   usage (through gpurun): bash tools/r4_small_batch_trace.sh 8
set -u
R=${1:-8}
out=gpurun_out/small_R$R
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d $out/t -o p -- python bench.py --restarts $R --steps 12 --warmup 2 --no-cpu-baseline --no-aux > $out/bench.json 2> $out/stderr.txt
db=$(find $out/t -name '*.db' | head -1)
python tools/rocpd_timeline.py $db 40 > $out/timeline.md
python tools/rocpd_summary.py $db > $out/kernel_trace.md
rm -rf $out/t
cat $out/timeline.md
