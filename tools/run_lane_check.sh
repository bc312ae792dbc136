# after a change to the lane-per-sample kernel: its tests, the class sweep against the matrix-core kernel, the per-wave timeline at configs[1]
# usage (through gpurun): bash tools/run_lane_check.sh <tag>
tag=${1:-lc}
out=gpurun_out/$tag
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_lane.py tests/test_gpu_elbo.py tests/test_gpu_parity_blocks.py -m gpu -x -q 2>&1 | tail -5 > $out/pytest.txt
cat $out/pytest.txt
VBMC_DEBUG_OCC=1 python tools/small_probe.py 6 200 10 1000 8 64 2>&1 | grep "lane deal\|^{" | sort -u | tail -3 | tee $out/probe_c1.txt
bash tools/run_lane_sweep.sh 2>&1 | tee $out/sweep.txt
if [ -f vbmc_amd/lib/tune/lib_inst.so ]; then
  VBMC_HIP_LIB=vbmc_amd/lib/tune/lib_inst.so python tools/lane_timeline.py 6 200 10 1000 8 64 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tee $out/timeline_c1.txt
fi
