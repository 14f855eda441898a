set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r3_tests_l.log 2>&1
rc=$?; echo "all tests rc=$rc"; tail -4 gpurun_out/r3_tests_l.log
if [ $rc -ne 0 ]; then exit $rc; fi
one() { python tools/bench_kernels.py "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g M/s  %.4f ms' % (d['units_per_s']/1e6, d['ms_mean']))"; }
for rep in 1 2; do
echo -n "f32 solve    "; one --mode solve --config cfg4
echo -n "f32 solve_pc "; one --mode solve_pc --config cfg4
echo -n "f32 step     "; one --mode step --config cfg4
echo -n "f64 step     "; one --mode step --config cfg3
echo -n "f64 solve    "; one --mode solve --config cfg3
done
