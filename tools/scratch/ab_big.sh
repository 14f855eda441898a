set -e
for lib in tools/ab_libs/libminiopt_rpt2.so ""; do
  echo "== lib=${lib:-product(rpt4)}"
  MO_LIB_PATH=$lib timeout -k 10 200 python tools/bench_kernels.py --mode step --shape 512,40,128,600 --batch 512 --reps 3 --warmup 1 2>&1 | tail -1 | cut -c1-200
  MO_LIB_PATH=$lib timeout -k 10 200 python tools/bench_kernels.py --mode step --shape 384,32,96,400 --batch 1024 --reps 3 --warmup 1 2>&1 | tail -1 | cut -c1-200
  MO_LIB_PATH=$lib timeout -k 10 120 python tools/bench_kernels.py --mode step --shape 256,40,128,300 --batch 2048 --reps 5 --warmup 1 2>&1 | tail -1 | cut -c1-200
done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "beyond or large" 2>&1 | tail -2
