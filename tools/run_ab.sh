# A/B table of tools/ent_ab.py on the GPU box: bash tools/run_ab.sh <tag> "<ENT_AB spec>" [reps]
set -u
tag=${1:-ab}
export ENT_AB="${2:-}"
mkdir -p gpurun_out/$tag
ENT_AB_REPS=${3:-3} timeout 1500 python tools/ent_ab.py run 64 10000 > gpurun_out/$tag/ab.txt 2>&1
cat gpurun_out/$tag/ab.txt
