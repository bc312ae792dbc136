# round-5 A/B run on the GPU box: entropy parity tests against the tree's library, the variant table of tools/ent_ab.py, one bench line
set -u
tag=${1:-r5a}
mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests/test_gpu_elbo.py tests/test_gpu_known_answers.py tests/test_gpu_fullsize.py tests/test_gpu_parity_blocks.py tests/test_gpu_random_shapes.py tests/test_gpu_limits.py tests/test_gpu_shard_s.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/$tag/pytest.txt
ENT_AB_REPS=${REPS:-4} timeout 1500 python tools/ent_ab.py run 64 10000 > gpurun_out/$tag/ab.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/$tag/bench_err.txt | grep '^{' | tail -1 > gpurun_out/$tag/bench.json
cat gpurun_out/$tag/pytest.txt gpurun_out/$tag/ab.txt
