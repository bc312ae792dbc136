set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for ns in 28 10000; do
  out=gpurun_out/adam_ns$ns
  rm -rf $out; mkdir -p $out
  PROF_NS=$ns rocprofv3 --kernel-trace -d $out/t -o p -- python tools/prof_adam.py > $out/run.txt 2>&1
  db=$(find $out/t -name '*.db' | head -1)
  python tools/rocpd_summary.py $db > $out/summary.md
  python tools/rocpd_timeline.py $db 16 > $out/timeline.md
  rm -rf $out/t
  echo "=== Ns=$ns"; grep "us/iter" $out/run.txt; head -12 $out/summary.md | cut -c1-160; cat $out/timeline.md | cut -c1-120
done
