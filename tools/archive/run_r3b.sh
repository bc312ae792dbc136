set -u
out=gpurun_out/r3b
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -X faulthandler -m pytest tests/test_gpu_mex.py -m gpu -q 2>&1 | grep -v "^Extension modules" | tail -120 > $out/test_gpu_mex.txt
echo "== mex: $(tail -1 $out/test_gpu_mex.txt)"
