# lane-per-sample kernel against the matrix-core kernel over the small class: pipelined rate and kernel time (tools/small_probe.py)
# usage: bash tools/run_lane_sweep.sh ["D N K Ns S R" ...]
if [ $# -eq 0 ]; then set -- "2 30 2 100 1 1" "2 30 2 100 1 64" "6 200 10 1000 8 1" "6 200 10 1000 8 8" "6 200 10 1000 8 64" "6 200 10 100 8 1" "6 200 10 100 8 64" "4 100 4 200 4 16" "12 300 16 1000 8 64" "10 300 12 1000 8 64" "10 300 10 1000 8 64" "12 300 8 1000 8 64" "8 200 16 2000 8 32" "3 60 6 10000 4 64" "6 200 10 10000 8 64"; fi
for shape in "$@"; do
  for e in "X=1" "VBMC_ENT_KERNEL=mfma"; do
    env $e python tools/small_probe.py $shape 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('%-28s %-22s %10.0f evals/s  step %7.1f us  kernel %7.1f us  lj %6.1f us  blocking %7.1f us' % (' '.join(str(x) for x in d['shape']), d['env'].get('VBMC_ENT_KERNEL', 'lane'), d['evals_per_s'], d['us_per_step'], d['ent_kernel_us_med_min'][0], d['lj_kernel_us'], d['blocking_us']))
"
  done
done
