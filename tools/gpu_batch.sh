set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "two_y_tiles or decrease_mu or up_to_128_variables" > gpurun_out/r3_tests_i.log 2>&1
rc=$?; echo "ny2 tests rc=$rc"; tail -4 gpurun_out/r3_tests_i.log
if [ $rc -ne 0 ]; then exit $rc; fi
MO_NY2_SOLVE_WPS=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "two_y_tiles or decrease_mu" > gpurun_out/r3_tests_i2.log 2>&1
rc=$?; echo "ny2 wps1 tests rc=$rc"; tail -4 gpurun_out/r3_tests_i2.log
if [ $rc -ne 0 ]; then exit $rc; fi
one() { python tools/bench_kernels.py "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g M/s  %.4f ms' % (d['units_per_s']/1e6, d['ms_mean']))"; }
for rep in 1 2; do
  echo -n "k24 solve two waves  "; one --mode solve --shape 64,24,32,128
  echo -n "k24 solve one wave   "; MO_NY2_SOLVE_WPS=1 one --mode solve --shape 64,24,32,128
  echo -n "k24 solve_pc two waves  "; one --mode solve_pc --shape 64,24,32,128
  echo -n "k24 solve_pc one wave   "; MO_NY2_SOLVE_WPS=1 one --mode solve_pc --shape 64,24,32,128
done
