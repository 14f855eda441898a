set -e
for i in 1 2; do
  timeout -k 10 120 python tools/bench_kernels.py --mode step --shape 256,40,128,300 --batch 2048 --reps 5 --warmup 1 2>&1 | tail -1 | cut -c1-200
  timeout -k 10 120 python tools/bench_kernels.py --mode step --shape 200,20,64,256 --batch 2048 --reps 5 --warmup 1 2>&1 | tail -1 | cut -c1-200
  timeout -k 10 120 python tools/bench_kernels.py --mode generic --config cfg3 --reps 5 --warmup 1 2>&1 | tail -1 | cut -c1-200
  timeout -k 10 120 python tools/bench_kernels.py --mode generic --config cfg2 --batch 65536 --reps 5 --warmup 1 2>&1 | tail -1 | cut -c1-200
done
timeout -k 10 120 python tools/bench_kernels.py --mode solve --shape 256,40,128,300 --batch 2048 --reps 3 --warmup 1 2>&1 | tail -1 | cut -c1-330
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_nls.py -x -q -m gpu 2>&1 | tail -2
