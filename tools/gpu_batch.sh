set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r3_tests_d.log 2>&1
rc=$?; echo "all tests rc=$rc"; tail -8 gpurun_out/r3_tests_d.log
if [ $rc -ge 124 ]; then exit $rc; fi
python tools/bench_kernels.py --mode linearize --config cfg4 > gpurun_out/r3_lin_f32.jsonl 2>gpurun_out/r3_lin_f32.err || exit 1
cat gpurun_out/r3_lin_f32.jsonl
python tools/bench_kernels.py --mode linearize --config cfg3 
python3 tools/profile.py r03_linearize_f32_cfg4 --batch 65536 -- tools/bench_kernels.py --mode linearize --config cfg4 || exit 1
python - <<'PY'
import json
for t in ("r03_linearize_f32_cfg4",):
    d=json.load(open(f"gpurun_out/prof_{t}/summary.json"))
    for k,v in d.items():
        if k!="_meta": print(t,k[:60],{x:v.get(x) for x in ("FETCH_SIZE","WRITE_SIZE","SQ_INSTS_VALU","SQ_INSTS_MFMA")}, v.get("kernel_stats"))
PY
