set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r3_tests_j.log 2>&1
rc=$?; echo "all tests rc=$rc"; tail -4 gpurun_out/r3_tests_j.log
if [ $rc -ne 0 ]; then exit $rc; fi
for rep in 1 2 3; do
  MO_FUSED_WPS=3 python bench.py --config cfg2 --no-cpu-baseline --sustain-seconds 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 3 waves', d['value'], d['roofline']['kernel_ms'])"
  python bench.py --config cfg2 --no-cpu-baseline --sustain-seconds 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 auto   ', d['value'], d['roofline']['kernel_ms'])"
done
python bench.py --config cfg2 --sustain-seconds 0.5 > gpurun_out/r3_bench_cfg2.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r3_bench_cfg2.json')); print('cfg2', d['value'], d['roofline']['frac'], d['parity'], d['cpu_baseline']['value'], d['cpu_baseline']['one_core']['value'])"
python bench.py --config cfg4 --steps 20 --warmup 5 --sustain-seconds 0.5 > gpurun_out/r3_bench_cfg4.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r3_bench_cfg4.json')); print('cfg4', d['value'], d['roofline']['frac'], d['parity'], d['cpu_baseline']['value'], d['cpu_baseline']['one_core']['value'])"
