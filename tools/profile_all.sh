#!/bin/bash
# Re-collect every profile tag that profiles/README.md quotes (round 4).  Run on the GPU box from the repository root:
#   MO_GIT_HEAD=<short hash> bash tools/profile_all.sh [a|b|c]      (no argument: all parts; each part fits one gpurun call)
# then `python3 tools/collect_profiles.py r04` copies gpurun_out/prof_<tag>/{summary.json,kernel_stats.csv} to
# profiles/<tag>_{pmc_summary.json,kernel_stats.csv}.  The step and Solve tags that bench.py quotes its `roofline.traffic` from are
# collected THROUGH bench.py (same workload, same batch); the others through tools/bench_kernels.py.
set -e
P="python3 tools/profile.py"
R=r04
B="--no-cpu-baseline --sustain-seconds 0"
if [ -z "$1" ] || [ "$1" = "a" ]; then
  $P ${R}_step_cfg3 --batch 65536 -- bench.py $B
  $P ${R}_step_cfg5shard --batch 131072 -- bench.py $B --config cfg5 --batch 131072
  $P ${R}_step_cfg2 --batch 4096 -- bench.py $B --config cfg2
  $P ${R}_f32_cfg4 --batch 65536 -- bench.py $B --config cfg4 --steps 20 --warmup 5
  $P ${R}_solve_cfg3 --batch 65536 -- bench.py $B --mode solve --steps 10 --warmup 3
  $P ${R}_solve_pc_cfg3 --batch 65536 -- bench.py $B --mode solve --strategy pc --steps 10 --warmup 3
  $P ${R}_solve_cfg2 --batch 65536 -- bench.py $B --mode solve --config cfg2 --batch 65536 --steps 10 --warmup 3
fi
if [ -z "$1" ] || [ "$1" = "b" ]; then
  $P ${R}_solve_k24 --batch 65536 -- tools/bench_kernels.py --mode solve --shape 64,24,32,128
  $P ${R}_solve_pc_k24 --batch 65536 -- tools/bench_kernels.py --mode solve_pc --shape 64,24,32,128
  $P ${R}_step_k24 --batch 65536 -- tools/bench_kernels.py --mode step --shape 64,24,32,128
  $P ${R}_generic_large_n256 --batch 2048 -- tools/bench_kernels.py --mode step --shape 256,40,128,300 --batch 2048 --reps 5 --warmup 1
  $P ${R}_generic_large_solve_n256 --batch 2048 -- tools/bench_kernels.py --mode solve --shape 256,40,128,300 --batch 2048 --reps 3 --warmup 1
  $P ${R}_generic_cfg3 --batch 65536 -- tools/bench_kernels.py --mode generic --config cfg3 --reps 5 --warmup 1
fi
if [ -z "$1" ] || [ "$1" = "c" ]; then
  $P ${R}_linearize_cfg3 --batch 65536 -- tools/bench_kernels.py --mode linearize --config cfg3
  $P ${R}_linearize_f32_cfg4 --batch 65536 -- tools/bench_kernels.py --mode linearize --config cfg4
  $P ${R}_step_cfg3_colmajor --batch 65536 -- tools/bench_kernels.py --mode step --config cfg3 --layout col
  $P ${R}_solve_f32_cfg4 --batch 65536 -- tools/bench_kernels.py --mode solve --config cfg4
  $P ${R}_solve_tiny --batch 65536 -- tools/bench_kernels.py --mode solve --shape 8,2,4,16
  $P ${R}_step_n128 --batch 16384 -- tools/bench_kernels.py --mode step --shape 128,14,64,256 --batch 16384
fi
