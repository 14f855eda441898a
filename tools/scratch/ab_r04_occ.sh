# LARGE path: two workgroups per CU (256 VGPRs, 4 x 4 tile quadrants) vs three (168 VGPRs, 3 x 3): tools/ab_libs/libminiopt_occ3.so
out=gpurun_out/ab_r04_occ.txt
mkdir -p gpurun_out; : > $out
for i in 1 2; do
for lib in "" tools/ab_libs/libminiopt_occ3.so; do
  for args in "--mode generic --config cfg3 --reps 5 --warmup 1" "--mode step --shape 90,40,50,180 --batch 16384 --reps 5 --warmup 1" "--mode solve --shape 90,40,50,180 --batch 16384 --reps 3 --warmup 1" "--mode step --shape 128,40,64,256 --batch 8192 --reps 5 --warmup 1" "--mode step --shape 160,16,32,170 --batch 4096 --reps 5 --warmup 1" "--mode step --shape 200,20,64,256 --batch 2048 --reps 5 --warmup 1" "--mode step --shape 256,40,128,300 --batch 2048 --reps 5 --warmup 1"; do
    echo "== lib=${lib:-product} $args" >> $out
    MO_LIB_PATH=$lib timeout -k 10 200 python tools/bench_kernels.py $args 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('kernel','ms_mean','units_per_s','mean_iterations') if k in d})" >> $out
  done
done
done
cat $out
