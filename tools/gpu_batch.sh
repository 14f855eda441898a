set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "beyond_every_lds or large_fp64" > gpurun_out/large_tests.log 2>&1
B="tools/bench_kernels.py --mode step --shape 256,40,128,300 --batch 2048 --reps 5 --warmup 1"
python $B > gpurun_out/large_step.log 2>&1
MO_LIB_PATH=tools/ab_libs/libminiopt_nopair.so python $B > gpurun_out/large_step_nopair.log 2>&1
B2="tools/bench_kernels.py --mode solve --shape 256,40,128,300 --batch 2048 --reps 3 --warmup 1"
python $B2 > gpurun_out/large_solve.log 2>&1
MO_LIB_PATH=tools/ab_libs/libminiopt_nopair.so python $B2 > gpurun_out/large_solve_nopair.log 2>&1
