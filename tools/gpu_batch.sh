set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1
python tools/bench_f32_offgrid.py > gpurun_out/f32_offgrid.log 2>&1
python tools/bench_f32_offgrid.py 36 4 20 64 65536 >> gpurun_out/f32_offgrid.log 2>&1
