set -u
out=gpurun_out/r3a
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for t in test_gpu_mex test_gpu_comm test_gpu_pipeline test_gpu_cold_build; do
  timeout 600 python -X faulthandler -m pytest tests/$t.py -m gpu -q -x 2>&1 | grep -v "^Extension modules" | tail -60 > $out/$t.txt
  echo "== $t: $(tail -1 $out/$t.txt)"
done
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -x --deselect tests/test_gpu_mex.py --deselect tests/test_gpu_comm.py 2>&1 | grep -v "^Extension modules" | tail -40 > $out/pytest_gpu.txt
tail -3 $out/pytest_gpu.txt
