set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "three_and_four or two_y_tiles" > gpurun_out/r3_tests_n.log 2>&1
rc=$?; echo "ny34 tests rc=$rc"; tail -30 gpurun_out/r3_tests_n.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 700 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r3_tests_n2.log 2>&1
rc=$?; echo "all tests rc=$rc"; tail -5 gpurun_out/r3_tests_n2.log
one() { python tools/bench_kernels.py "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel'], '%.4g M/s  %.4f ms' % (d['units_per_s']/1e6, d['ms_mean']))"; }
echo -n "k40 step    "; one --mode step --shape 64,40,32,128
echo -n "k40 generic "; one --mode generic --shape 64,40,32,128 --reps 3 --warmup 1
echo -n "k40 solve   "; one --mode solve --shape 64,40,32,128
echo -n "k56 step    "; one --mode step --shape 64,56,32,128
echo -n "k56 solve   "; one --mode solve --shape 64,56,32,128
