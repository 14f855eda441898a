set -e
for i in 1 2; do
  timeout -k 10 120 python tools/bench_kernels.py --mode step --shape 256,40,128,300 --batch 2048 --reps 5 --warmup 1 2>&1 | tail -1 | cut -c1-200
  timeout -k 10 120 python tools/bench_kernels.py --mode step --shape 200,20,64,256 --batch 2048 --reps 5 --warmup 1 2>&1 | tail -1 | cut -c1-200
  timeout -k 10 120 python tools/bench_kernels.py --mode step --shape 130,70,16,140 --batch 2048 --reps 5 --warmup 1 2>&1 | tail -1 | cut -c1-200
done
timeout -k 10 120 python tools/bench_kernels.py --mode solve --shape 256,40,128,300 --batch 2048 --reps 3 --warmup 1 2>&1 | tail -1 | cut -c1-330
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "beyond or large_fp64 or first_iteration_is_iterate" 2>&1 | tail -2
