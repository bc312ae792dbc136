#!/bin/bash
# The N > 1 code path of bench.py with ONE rank under torch.distributed.run (GPU box): weak leg and strong leg, per environment setting.
#   usage: bash tools/r4_forced_comm.sh R "NAME ENV=v ENV=v" ...
R=${1:-64}; shift
port=29520
for spec in "$@"; do
  set -- $spec
  name=$1; shift
  port=$((port + 1))
  env "$@" VBMC_BENCH_FORCE_COMM=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port \
    bench.py --gpus 1 --restarts $R --steps ${STEPS:-40} --no-aux --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'weak_ms', round(d['ms_per_step'],4), 'strong_ms', round(d['strong']['ms_per_step'],4) if d.get('strong') else None)"
done
