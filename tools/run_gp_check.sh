#!/bin/bash
# Correctness of the GP side after a change to the factorisation path: the stand-alone Cholesky harness (factor, by-products,
# failure index) over sizes, then the GP test files.   usage (through gpurun):  bash tools/run_gp_check.sh
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
(for v in 1 2; do for n in 5 16 37 100 250 400 576 577 592 800 1120 1121 1300; do timeout 120 vbmc_amd/lib/chol_bench $v $n 3 3 0 | cut -c1-230; done; done) 2>&1 | tee gpurun_out/chol_check.txt | grep -c '"ok": true'
grep -v '"ok": true' gpurun_out/chol_check.txt | head -20
timeout 1500 python -m pytest -q -x -m gpu tests/test_gpu_gplite.py tests/test_gpu_nlz.py tests/test_gpu_limits.py tests/test_gpu_acq.py tests/test_gpu_random_shapes.py tests/test_gpu_known_answers.py tests/test_gpu_system.py 2>&1 | tail -15
