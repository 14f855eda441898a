# A/B of the generic kernels (LDS-resident and LARGE): tools/ab_libs/libminiopt_r04c.so vs the product
out=gpurun_out/ab_r04_generic.txt
mkdir -p gpurun_out; : > $out
for i in 1 2; do
for lib in ${AB_LIBS:-tools/ab_libs/libminiopt_r04c.so} ""; do
  for args in "--mode generic --config cfg3 --reps 5 --warmup 1" "--mode generic --config cfg2 --reps 5 --warmup 1" "--mode step --shape 100,40,64,128 --batch 16384 --reps 5 --warmup 1" "--mode solve --shape 100,40,64,128 --batch 16384 --reps 3 --warmup 1" "--mode step --shape 256,40,128,300 --batch 2048" "--mode solve --shape 256,40,128,300 --batch 2048 --reps 3 --warmup 1"; do
    echo "== lib=${lib:-product} $args" >> $out
    MO_LIB_PATH=$lib timeout -k 10 200 python tools/bench_kernels.py $args 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('kernel','ms_mean','units_per_s','mean_iterations') if k in d})" >> $out
  done
done
done
cat $out
