# round-4 profile sweep, part a: the Solve kernels (bench.py --mode solve is the driver-timeable program; k = 24 through tools/bench_kernels.py)
set -e
P="python3 tools/profile.py"
R=r04
$P ${R}_solve_cfg3 --batch 65536 -- bench.py --mode solve --no-cpu-baseline --sustain-seconds 0 --steps 20 --warmup 5
$P ${R}_solve_pc_cfg3 --batch 65536 -- bench.py --mode solve --strategy pc --no-cpu-baseline --sustain-seconds 0 --steps 20 --warmup 5
$P ${R}_solve_k24 --batch 65536 -- tools/bench_kernels.py --mode solve --shape 64,24,32,128
