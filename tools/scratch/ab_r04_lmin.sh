# LDS-resident generic kernel vs the LARGE path on shapes both can run (tools/ab_libs/libminiopt_lmin16.so: -DMO_LARGE_MIN_P=16)
out=gpurun_out/ab_r04_lmin.txt
mkdir -p gpurun_out; : > $out
for i in 1 2; do
for lib in "" ${AB_LIBS:-tools/ab_libs/libminiopt_lmin16.so}; do
  for args in "--mode generic --config cfg2 --reps 5 --warmup 1" "--mode generic --config cfg2 --batch 65536 --reps 5 --warmup 1" "--mode generic --shape 40,8,16,64 --batch 65536 --reps 5 --warmup 1" "--mode generic --shape 50,6,16,64 --batch 65536 --reps 5 --warmup 1" "--mode generic --shape 56,8,16,64 --batch 65536 --reps 5 --warmup 1" "--mode generic --config cfg3 --reps 5 --warmup 1" "--mode generic --shape 16,4,8,32 --batch 65536 --reps 5 --warmup 1"; do
    echo "== lib=${lib:-product} $args" >> $out
    MO_LIB_PATH=$lib timeout -k 10 200 python tools/bench_kernels.py $args 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('kernel','ms_mean','units_per_s','mean_iterations') if k in d})" >> $out
  done
done
done
cat $out
