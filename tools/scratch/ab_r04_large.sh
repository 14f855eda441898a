# A/B of the LARGE generic path on one box: tools/ab_libs/libminiopt_r04a.so (right-looking, before the left-looking factorisation) vs the product
out=gpurun_out/ab_r04_large.txt
mkdir -p gpurun_out; : > $out
for i in 1 2; do
for lib in ${AB_LIBS:-tools/ab_libs/libminiopt_r04a.so} ""; do
  for args in "--mode step --shape 256,40,128,300 --batch 2048" "--mode solve --shape 256,40,128,300 --batch 2048 --reps 3 --warmup 1" "--mode step --shape 200,20,64,256 --batch 2048" "--mode step --shape 160,16,32,170 --batch 2048" "--mode step --shape 384,32,96,400 --batch 1024 --reps 3 --warmup 1" "--mode step --shape 512,40,128,600 --batch 512 --reps 3 --warmup 1"; do
    echo "== lib=${lib:-product} $args" >> $out
    MO_LIB_PATH=$lib timeout -k 10 200 python tools/bench_kernels.py $args 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('kernel','ms_mean','units_per_s','mean_iterations','satisfied_frac') if k in d})" >> $out
  done
done
done
cat $out
