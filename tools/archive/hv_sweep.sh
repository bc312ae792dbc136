# entropy kernel at 64 < K <= 128: two waves (KT <= 4 each) or four waves (KT <= 2 each) per workgroup?  A = libA.so, B = libB.so
for cfg in "--D 20 --N 800 --K 100 --Ns 20000 --S 20 --steps 5" "--D 10 --N 400 --K 100 --Ns 10000 --S 20 --steps 5 --restarts 32" "--D 6 --N 200 --K 80 --Ns 8000 --S 8 --steps 5 --restarts 64" "--D 14 --N 300 --K 128 --Ns 8000 --S 8 --steps 5 --restarts 32" "--D 28 --N 300 --K 96 --Ns 8000 --S 8 --steps 5 --restarts 32"; do
  for hv in 2 4; do for v in ${LIBS:-B}; do
    VBMC_ENT_HV=$hv VBMC_HIP_LIB=$PWD/vbmc_amd/lib/lib$v.so python bench.py --no-aux --no-cpu-baseline $cfg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v HV=$hv', '$cfg'[:24], d['roofline']['kernel'], round(d['value'],1), round(d['roofline']['kernel_ms'],3))"
  done; done
done
