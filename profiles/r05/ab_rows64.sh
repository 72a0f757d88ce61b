cd $GRAFT_REPO_ROOT
for pass in 1 2 3; do
 for v in 0 1; do
  r=$(VDS_ROWS64=$v VDS_LIB=$PWD/build/libvds_r64.so timeout 200 python bench.py --steps 300 --warmup 3 --no-cpu-baseline --no-neighbour-leg --distinct-days 0 --no-hooked-leg --no-fallbacks-leg --check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/day  one chain %.3f ms  check %s' % (d['ms_per_step'], d['roofline']['one_chain_day_kernel_ms'], d.get('parity_check_vs_oracle')))")
  echo "pass $pass rows64=$v $r"
 done
done
