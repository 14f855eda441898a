set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r3_tests_k.log 2>&1
rc=$?; echo "all tests rc=$rc"; tail -4 gpurun_out/r3_tests_k.log
if [ $rc -ne 0 ]; then exit $rc; fi
for rep in 1 2 3; do
  python bench.py --config cfg2 --no-cpu-baseline --sustain-seconds 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 ', d['value'], d['roofline']['kernel_ms'])"
  python bench.py --no-cpu-baseline --sustain-seconds 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 ', d['value'], d['roofline']['kernel_ms'])"
done
one() { python tools/bench_kernels.py "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g M/s  %.4f ms' % (d['units_per_s']/1e6, d['ms_mean']))"; }
echo -n "solve    "; one --mode solve --config cfg3
echo -n "tiny solve "; one --mode solve --shape 8,2,4,16
echo -n "tiny step "; one --mode step --shape 8,2,4,16
echo -n "cfg2 step 65536 "; one --mode step --config cfg2 --batch 65536
