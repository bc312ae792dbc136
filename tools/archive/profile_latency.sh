#!/bin/bash
# The latency-bound paths of round 3 on the MI355X box: one Adam chain on the device (tools/prof_adam.py) with the round-2
# schedule (VBMC_FIN=seq) and the round-3 one, at VBMC's own sample count and at Ns = 1e4, and the GP entry points for few
# matrices (tools/prof_gp_post.py) -- un-profiled rates, then rocprofv3 kernel traces.  Writes gpurun_out/latency/{adam,gp}.md.
#   usage (through gpurun):  bash tools/profile_latency.sh
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/latency; rm -rf $out; mkdir -p $out
{
echo "# One Adam chain on the device (vbmc_adam_batch, R = 1) at D = 10, N = 400, K = 50, S = 20 -- tools/profile_latency.sh"
echo
echo "Un-profiled, 200 iterations after 40 warm-up iterations (\`tools/prof_adam.py\`):"
echo
echo "| schedule | Ns per component | us per iteration |"
echo "|---|---|---|"
for f in seq:0 ws:0 ws:1; do for ns in 28 10000; do
  v=$(VBMC_FIN=${f%:*} VBMC_LJ_CO=${f#*:} PROF_NS=$ns timeout 120 python tools/prof_adam.py 2>&1 | tail -1 | sed 's/us\/iter //')
  case $f in
    seq:0) name="round 2 (\`VBMC_FIN=seq VBMC_LJ_CO=0\`: k_prep+Adam, k_logjoint, entropy, k_reduce_both, k_finalize)";;
    ws:0) name="round 3, first half (\`VBMC_LJ_CO=0\`: k_logjoint, entropy, k_reduce_both, k_finalize_ws + Adam + unpacking)";;
    *) name="round 3 (entropy launch with the log-joint role, k_reduce_both, k_finalize_ws + Adam + unpacking)";;
  esac
  echo "| $name | $ns | $v |"
done; done
for f in seq:0 ws:0 ws:1; do for ns in 28 10000; do
  VBMC_FIN=${f%:*} VBMC_LJ_CO=${f#*:} PROF_NS=$ns rocprofv3 --kernel-trace -d $out/t -o p -- python tools/prof_adam.py > /dev/null 2>&1
  db=$(find $out/t -name '*.db' | head -1)
  echo; echo "## Kernel trace, VBMC_FIN=${f%:*} VBMC_LJ_CO=${f#*:}, Ns = $ns (rocprofv3 --kernel-trace; the profiler adds ~5-8 us per iteration to the wall time)"; echo
  python tools/rocpd_summary.py $db | head -9 | cut -c1-170
  echo; echo "Last iterations:"; echo
  python tools/rocpd_timeline.py $db 16 | cut -c1-120 | head -14
  rm -rf $out/t
done; done
} > $out/adam.md
{
echo "# GP entry points for few matrices at N = 400, D = 10 -- tools/profile_latency.sh"
echo
echo "\`tools/prof_gp_post.py\`: gplite_post with the factors left on the device (S = 20), gplite_nlZ + gradient for one hyper-parameter vector:"
echo
echo '```'
python tools/prof_gp_post.py
CALLS=12 NEED_L=1 python tools/gp_post_probe.py 2>/dev/null | tail -1
echo '```'
rocprofv3 --kernel-trace -d $out/t -o p -- python tools/prof_gp_post.py > /dev/null 2>&1
db=$(find $out/t -name '*.db' | head -1)
echo; echo "## Kernel trace of the same script"; echo
python tools/rocpd_summary.py $db | head -18 | cut -c1-170
echo; echo "Timeline of the last gplite_nlZ calls (one call = k_gp_scale ... k_nlz_final):"; echo
python tools/rocpd_timeline.py $db 34 | cut -c1-120
rm -rf $out/t
} > $out/gp.md
head -12 $out/adam.md; sed -n 1,12p $out/gp.md
