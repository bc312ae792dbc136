# A/B of two builds of the library on one box: put them at vbmc_amd/lib/libA.so and libB.so (git stash / build / cp), then
#   gpurun -- 'BENCH_ARGS="--D 20 --K 100 ..." bash tools/ab_bench.sh'     -> alternating bench runs: "A|B evals/s kernel_ms"
for i in 1 2 3; do for v in A B; do VBMC_HIP_LIB=$PWD/vbmc_amd/lib/lib$v.so python bench.py --no-aux --no-cpu-baseline ${BENCH_ARGS:-} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['roofline']['kernel_ms'])"; done; done
