# A/B of the fp32 step kernel at BASELINE configs[3]: right-hand side as a vector (tools/ab_libs/libminiopt_f32rv.so) vs tile column NT + 1 (product)
out=gpurun_out/ab_r04_f32.txt
mkdir -p gpurun_out; : > $out
for i in 1 2 3; do
for lib in tools/ab_libs/libminiopt_f32rv.so ""; do
  echo "== lib=${lib:-product}" >> $out
  MO_LIB_PATH=$lib timeout -k 10 200 python tools/bench_kernels.py --mode step --config cfg4 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('kernel','ms_mean','units_per_s') if k in d})" >> $out
done
done
MO_LIB_PATH=tools/ab_libs/libminiopt_f32rv.so timeout -k 10 600 python -m pytest tests -m gpu -q -k "fp32 or f32" 2>&1 | tail -5 >> $out
cat $out
