set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1
python bench.py > gpurun_out/bench_default.log 2>&1
python bench.py --config cfg2 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_cfg2.log 2>&1
