set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1
for i in 1 2; do
python bench.py --config cfg2 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_cfg2_$i.log 2>&1
python bench.py --config cfg2 --batch 1024 --steps 200 --warmup 20 --no-cpu-baseline --sustain-seconds 0 > gpurun_out/bench_cfg2_1k_$i.log 2>&1
python bench.py --config cfg3 --batch 4096 --steps 200 --warmup 20 --no-cpu-baseline --sustain-seconds 0 > gpurun_out/bench_cfg3_4k_$i.log 2>&1
done
