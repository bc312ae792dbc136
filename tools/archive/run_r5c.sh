# round 5: the whole GPU suite (relative relerr), the A/B table with the kernel timed alone, one bench line
set -u
tag=${1:-r5c}
mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -40 > gpurun_out/$tag/pytest.txt
ENT_AB_REPS=${REPS:-4} timeout 1500 python tools/ent_ab.py run 64 10000 > gpurun_out/$tag/ab.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/$tag/bench_err.txt | grep '^{' | tail -1 > gpurun_out/$tag/bench.json
tail -30 gpurun_out/$tag/pytest.txt; cat gpurun_out/$tag/ab.txt
