#!/bin/bash
# GP entry points for few matrices (gplite_post resident S = 20, gplite_nlZ + gradient B = 1) at N = 400, D = 10:
# un-profiled wall times, then a kernel + HIP API + copy trace of the same script.  Writes gpurun_out/gplat_<tag>/.
#   usage (through gpurun):  bash tools/run_gp_lat.sh <tag>
set -u
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/gplat_$tag; rm -rf $out; mkdir -p $out
python tools/prof_gp_post.py > $out/wall.txt 2>&1
python tools/prof_gp_post.py >> $out/wall.txt 2>&1
cat $out/wall.txt
if [ "${GPLAT_TRACE:-1}" = 1 ]; then
rocprofv3 --kernel-trace --hip-trace --memory-copy-trace -d /tmp/gpl -o t -- python tools/prof_gp_post.py > /dev/null 2>&1
db=$(find /tmp/gpl -name '*results.db' | head -1)
python tools/rocpd_summary.py $db | head -30 | cut -c1-170 > $out/kernels.md
python tools/rocpd_timeline.py $db 60 > $out/timeline.md
python tools/rocpd_api_summary.py $db 32 > $out/api.md
rm -rf /tmp/gpl
fi
[ -x vbmc_amd/lib/chol_bench ] && (timeout 60 vbmc_amd/lib/chol_bench 1 400 1 20 0; timeout 60 vbmc_amd/lib/chol_bench 1 400 20 20 0; timeout 60 vbmc_amd/lib/chol_bench 2 400 1 20 1; timeout 60 vbmc_amd/lib/chol_bench 2 400 20 20 0) > $out/chol_bench.txt 2>&1
tail -3 $out/chol_bench.txt 2>/dev/null | cut -c1-160
