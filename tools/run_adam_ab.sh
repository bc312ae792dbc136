# single-chain Adam (tools/prof_adam.py, VBMC's own Ns = 28 and Ns = 1e4) under two builds of the library, interleaved: bash tools/run_adam_ab.sh <libA> <libB>
set -u
for rep in 1 2 3; do
  for lib in "$@"; do
    for ns in 28 10000; do
      printf "%s Ns=%s " "$(basename $lib)" $ns
      VBMC_HIP_LIB=$lib PROF_NS=$ns python tools/prof_adam.py 2>/dev/null | tail -1
    done
  done
done
