set -e
for i in 1 2; do
for lib in "" tools/ab_libs/libminiopt_se16.so; do
  echo "== lib=${lib:-product(se8)}"
  MO_LIB_PATH=$lib timeout -k 10 120 python tools/bench_kernels.py --mode step --shape 256,40,128,300 --batch 2048 --reps 5 --warmup 1 2>&1 | tail -1 | cut -c90-190
  MO_LIB_PATH=$lib timeout -k 10 120 python tools/bench_kernels.py --mode step --shape 200,20,64,256 --batch 2048 --reps 5 --warmup 1 2>&1 | tail -1 | cut -c90-190
  MO_LIB_PATH=$lib timeout -k 10 120 python tools/bench_kernels.py --mode solve --shape 256,40,128,300 --batch 2048 --reps 3 --warmup 1 2>&1 | tail -1 | cut -c90-200
done
done
MO_LIB_PATH=tools/ab_libs/libminiopt_se16.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "beyond or large" 2>&1 | tail -2
