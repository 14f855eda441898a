set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 120 python tools/dbg_f32.py > gpurun_out/r3_dbg_f32.txt 2>&1; tail -40 gpurun_out/r3_dbg_f32.txt
timeout -k 10 120 ./tools/phase_timer 64 65536 > gpurun_out/r3_phase_timer_cfg3.txt 2>&1 || exit 1
cat gpurun_out/r3_phase_timer_cfg3.txt
for rep in 1 2 3; do
  python tools/bench_kernels.py --mode solve --config cfg3 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('product solve', d['units_per_s'], d['ms_mean'])"
  MO_LIB_PATH=$PWD/tools/ab_libs/libminiopt_jnt.so python tools/bench_kernels.py --mode solve --config cfg3 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('jnt     solve', d['units_per_s'], d['ms_mean'])"
done
python tools/bench_kernels.py --mode step --config cfg3 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('product step', d['units_per_s'], d['ms_mean'])"
MO_LIB_PATH=$PWD/tools/ab_libs/libminiopt_jnt.so python tools/bench_kernels.py --mode step --config cfg3 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('jnt     step', d['units_per_s'], d['ms_mean'])"
export MO_LIB_PATH=$PWD/tools/ab_libs/libminiopt_jnt.so
python3 tools/profile.py r03_solve_cfg3_jnt --batch 65536 -- tools/bench_kernels.py --mode solve --config cfg3 || exit 1
python - <<'PY'
import json
for t in ("r03_solve_cfg3_jnt",):
    d=json.load(open(f"gpurun_out/prof_{t}/summary.json"))
    for k,v in d.items():
        if k!="_meta": print(t,k[:60],{x:v.get(x) for x in ("FETCH_SIZE","WRITE_SIZE","SQ_INSTS_VALU","SQ_INSTS_MFMA")}, v.get("kernel_stats"))
PY
