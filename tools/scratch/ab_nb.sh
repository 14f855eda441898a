for i in 1 2; do
for lib in "" tools/ab_libs/libminiopt_nb16.so; do
  echo "== lib=${lib:-product(nb32)}"
  for shp in 160,16,32,170 130,70,16,140 190,3,2,193 145,0,0,150; do
    MO_LIB_PATH=$lib timeout -k 10 120 python tools/bench_kernels.py --mode step --shape $shp --batch 2048 --reps 5 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$shp step %.3f ms' % d['ms_mean'], end=' | ')"
  done
  echo
done
done
MO_LIB_PATH=tools/ab_libs/libminiopt_nb16.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "beyond or large" 2>&1 | tail -1
