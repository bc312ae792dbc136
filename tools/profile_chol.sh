#!/bin/bash
# Evidence for profiles/r02_chol.md on the MI355X box: the stand-alone Cholesky harness over sizes, the phase stamps, the
# single-CU tile-stream bandwidth, the GP-side aux bench, gplite_post wall times and the kernel trace of gplite_nlZ.
# Writes gpurun_out/chol/; assemble with  python tools/compose_chol_profile.py .
#   usage (through gpurun):  bash tools/profile_chol.sh
set -u
out=gpurun_out/chol
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
(for n in 16 37 100 250 400 592 593 800 1200 1300 2000; do timeout 60 vbmc_amd/lib/chol_bench 1 $n 20 10 0; done
 timeout 60 vbmc_amd/lib/chol_bench 1 400 256 20 0
 timeout 60 vbmc_amd/lib/chol_bench 1 400 1 20 1
 timeout 60 vbmc_amd/lib/chol_bench 9 400 1) > $out/chol_bench.txt 2>&1
timeout 200 python tools/bench_aux.py > $out/bench_aux.json 2>&1
python tools/gp_post_probe.py > $out/gp_post.txt 2>&1
NEED_L=1 python tools/gp_post_probe.py >> $out/gp_post.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pn -o t -- python tools/prof_nlz.py > /dev/null 2>&1
python tools/rocpd_summary.py $(find /tmp/pn -name '*results.db' | head -1) > $out/nlz_trace.md
rm -rf /tmp/pn
grep '"N": 400' $out/chol_bench.txt | cut -c1-120; grep gplite_post $out/gp_post.txt
