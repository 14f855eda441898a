set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/stress_overconstrained.jsonl gpurun_out/nls_disagreements.jsonl
timeout -k 10 700 python -m pytest tests -m gpu -q -p no:cacheprovider -x --deselect tests/test_gpu_fuzz.py > gpurun_out/r3_tests_a.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r3_tests_a.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider -s > gpurun_out/r3_tests_fuzz.log 2>&1
rc=$?; echo "fuzz rc=$rc"; tail -5 gpurun_out/r3_tests_fuzz.log
if [ $rc -ge 124 ]; then exit $rc; fi
bash tools/alpha_dual_probe/run_all.sh > gpurun_out/r3_probe.log 2>&1 || exit 1
python bench.py --steps 50 --warmup 10 > gpurun_out/r3_bench_a.json 2> gpurun_out/r3_bench_a.err || exit 1
for mode in solve solve_pc; do
  python tools/bench_kernels.py --mode $mode --config cfg3 >> gpurun_out/r3_solve_a.jsonl 2>> gpurun_out/r3_solve_a.err || exit 1
done
python tools/bench_kernels.py --mode solve --config cfg2 --batch 65536 >> gpurun_out/r3_solve_a.jsonl 2>> gpurun_out/r3_solve_a.err
python tools/bench_kernels.py --mode solve --config cfg4 >> gpurun_out/r3_solve_a.jsonl 2>> gpurun_out/r3_solve_a.err
tail -c 1500 gpurun_out/r3_solve_a.jsonl
