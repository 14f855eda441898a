#!/bin/bash
# Re-collect every profile tag that profiles/README.md quotes (round 2).  Run on the GPU box from the repository root, in two halves to stay
# inside one gpurun call each:   MO_GIT_HEAD=<short hash> bash tools/profile_all.sh a   /   ... b
# then copy gpurun_out/prof_<tag>/{summary.json,kernel_stats.csv} to profiles/<tag>_{pmc_summary.json,kernel_stats.csv}.
set -e
P="python3 tools/profile.py"
if [ "$1" = "a" ]; then
  $P r02_step_cfg3 --batch 65536 -- bench.py --no-cpu-baseline
  $P r02_step_cfg5shard --batch 131072 -- bench.py --no-cpu-baseline --config cfg5 --batch 131072
  $P r02_step_cfg2 --batch 4096 -- bench.py --no-cpu-baseline --config cfg2
  $P r02_f32_cfg4 --batch 65536 -- bench.py --no-cpu-baseline --config cfg4 --steps 20 --warmup 5
  $P r02_solve_cfg3 --batch 65536 -- tools/bench_kernels.py --mode solve --config cfg3
  $P r02_solve_pc_cfg3 --batch 65536 -- tools/bench_kernels.py --mode solve_pc --config cfg3
  $P r02_solve_cfg2 --batch 65536 -- tools/bench_kernels.py --mode solve --config cfg2 --batch 65536
  $P r02_linearize_cfg3 --batch 65536 -- tools/bench_kernels.py --mode linearize --config cfg3
else
  $P r02_generic_cfg3 --batch 65536 -- tools/bench_kernels.py --mode generic --config cfg3 --reps 5 --warmup 1
  $P r02_step_cfg3_colmajor --batch 65536 -- tools/bench_kernels.py --mode step --config cfg3 --layout col
  $P r02_solve_f32_cfg4 --batch 65536 -- tools/bench_kernels.py --mode solve --config cfg4
  $P r02_solve_pc_f32_cfg4 --batch 65536 -- tools/bench_kernels.py --mode solve_pc --config cfg4
  $P r02_linearize_f32_cfg4 --batch 65536 -- tools/bench_kernels.py --mode linearize --config cfg4
  $P r02_step_k24 --batch 65536 -- tools/bench_kernels.py --mode step --shape 64,24,32,128
  $P r02_solve_k24 --batch 65536 -- tools/bench_kernels.py --mode solve --shape 64,24,32,128
  $P r02_solve_tiny --batch 65536 -- tools/bench_kernels.py --mode solve --shape 8,2,4,16
fi
