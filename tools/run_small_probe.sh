# A/B runs of tools/small_probe.py over environment settings (one line of JSON per run)
for e in "X=1" "VBMC_LJ_CO=0" "VBMC_ENT_CHUNKS=2" "VBMC_ENT_CHUNKS=4"; do env $e VBMC_DEBUG_OCC=1 python tools/small_probe.py $SHAPE 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" | sort -u | tail -3; done
