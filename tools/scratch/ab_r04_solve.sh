# A/B on one box: round-3 library (tools/ab_libs/libminiopt_r03.so) against the product build, alternating.  Output: gpurun_out/ab_r04_solve.txt
set -e
out=gpurun_out/ab_r04_solve.txt
mkdir -p gpurun_out; : > $out
for i in 1 2 3; do
for lib in tools/ab_libs/libminiopt_r03.so ""; do
  for args in "--mode solve --config cfg3" "--mode solve_pc --config cfg3" "--mode solve --shape 64,24,32,128" "--mode solve --config cfg2 --batch 65536" "--mode step --config cfg3"; do
    echo "== lib=${lib:-product} $args" >> $out
    MO_LIB_PATH=$lib timeout -k 10 120 python tools/bench_kernels.py $args 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('kernel','ms_mean','units_per_s','mean_iterations','satisfied_frac') if k in d})" >> $out
  done
done
done
cat $out
