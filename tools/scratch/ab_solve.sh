set -e
for i in 1 2 3; do
for lib in "" tools/ab_libs/libminiopt_early.so; do
  echo "== lib=${lib:-product}"
  MO_LIB_PATH=$lib timeout -k 10 120 python tools/bench_kernels.py --mode solve --config cfg3 2>&1 | tail -1 | cut -c90-200
  MO_LIB_PATH=$lib timeout -k 10 120 python tools/bench_kernels.py --mode solve_pc --config cfg3 2>&1 | tail -1 | cut -c90-200
done
done
