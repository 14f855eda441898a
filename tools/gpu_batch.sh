set -e
mkdir -p gpurun_out
python tools/debug_case.py > gpurun_out/debug_case_final.log 2>&1
python tools/debug_case2.py > gpurun_out/debug_case2_final.log 2>&1
python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1
MO_FUZZ_EXTRA_SEEDS=100-139 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/fuzz_soak.log 2>&1
python tools/bench_kernels.py --mode solve --shape 128,14,64,256 --batch 16384 > gpurun_out/n128_solve.log 2>&1
python tools/bench_kernels.py --mode linearize --shape 128,14,64,256 --batch 16384 > gpurun_out/n128_lin.log 2>&1
