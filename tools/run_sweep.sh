# shape sweeps of tools/tune_sweep.py over the variants under vbmc_amd/lib/tune: bash tools/run_sweep.sh <tag> [all | Ks [Ds]]
set -u
tag=${1:-sweep}
mkdir -p gpurun_out/$tag
if [ -n "${2:-}" ] && [ "$2" != "all" ]; then
  TUNE_NO_HV=1 TUNE_KS="$2" TUNE_DS="${3:-6,10,14,18,20,24,28,32}" timeout 1500 python tools/tune_sweep.py > gpurun_out/$tag/custom.txt 2>&1
  cat gpurun_out/$tag/custom.txt
  exit 0
fi
timeout 1500 python tools/tune_sweep.py small > gpurun_out/$tag/small.txt 2>&1
TUNE_NO_HV=1 timeout 1500 python tools/tune_sweep.py > gpurun_out/$tag/large.txt 2>&1
if [ "${2:-}" = "all" ]; then
  TUNE_NO_HV=1 TUNE_KS="20,36,52,72,100,104,144,160,208,224" TUNE_DS="2,6,10,14,18,20,24,28,32" timeout 1500 python tools/tune_sweep.py > gpurun_out/$tag/tails.txt 2>&1
fi
tail -3 gpurun_out/$tag/small.txt; tail -3 gpurun_out/$tag/large.txt
