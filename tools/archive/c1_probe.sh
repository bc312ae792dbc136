for v in "" "VBMC_LJ_CO_SR=4 VBMC_LJ_CO_KR=8" "VBMC_LJ_CO_SR=4 VBMC_LJ_CO_KR=8 VBMC_PREP_UP=0 VBMC_FIN_FOLD=0" "VBMC_LJ_CO_SR=4 VBMC_LJ_CO_KR=8 VBMC_ENT_CHUNKS=2" "VBMC_LJ_CO_SR=4 VBMC_LJ_CO_KR=8 VBMC_ENT_CHUNKS=4" "VBMC_LJ_KERNEL=valu"; do
  for rep in 1 2; do
    echo -n "$v : "; env $v python bench.py --D 6 --N 200 --K 10 --Ns 1000 --S 8 --no-cpu-baseline --no-aux --steps 200 --warmup 20 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(1e3*d['ms_per_step'],1), round(1e3*d['roofline']['kernel_ms'],1), d.get('logjoint_kernel_ms'))"
  done
done
