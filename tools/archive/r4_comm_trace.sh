#!/bin/bash
# Kernel / copy timeline of the pipelined step through a one-rank communicator (GPU box).   usage: bash tools/r4_comm_trace.sh R [ENV=v ...]
set -u
R=${1:-64}; shift
tag=${TAG:-comm}
out=gpurun_out/commtrace_${tag}_R$R
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
env "$@" PROBE_R=$R PROBE_ONLY=${ONLY:-comm} PROBE_REPS=1 rocprofv3 --kernel-trace --memory-copy-trace -d $out/t -o p -- python tools/r4_comm_probe.py one > $out/probe.json 2> $out/stderr.txt
db=$(find $out/t -name '*.db' | head -1)
python tools/rocpd_timeline.py $db ${NK:-90} > $out/timeline.md
rm -rf $out/t
tail -1 $out/probe.json | cut -c1-300
