set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r3_tests_f.log 2>&1
rc=$?; echo "all tests rc=$rc"; tail -4 gpurun_out/r3_tests_f.log
if [ $rc -ge 124 ]; then exit $rc; fi
one() { python tools/bench_kernels.py "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g M/s  %.4f ms' % (d['units_per_s']/1e6, d['ms_mean']))"; }
for rep in 1 2 3; do
 for v in product pad4 vrow vrow_pad4; do
  if [ $v = product ]; then unset MO_LIB_PATH; else export MO_LIB_PATH=$PWD/tools/ab_libs/libminiopt_$v.so; fi
  echo -n "$v solve    "; one --mode solve --config cfg3
  echo -n "$v solve_pc "; one --mode solve_pc --config cfg3
  echo -n "$v step     "; one --mode step --config cfg3
 done
done
