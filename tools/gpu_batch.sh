set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1
for sr in 0 8 32 -1; do
for b in 4096 16384 65536; do
for mode in solve solve_pc linearize; do
MO_FUSED_STATIC_ROUNDS=$sr python tools/bench_kernels.py --mode $mode --config cfg3 --batch $b > gpurun_out/srm${sr}_${mode}_cfg3_b$b.log 2>&1
MO_FUSED_STATIC_ROUNDS=$sr python tools/bench_kernels.py --mode $mode --config cfg2 --batch $b > gpurun_out/srm${sr}_${mode}_cfg2_b$b.log 2>&1
done
done
for b in 4096 65536; do
MO_FUSED_STATIC_ROUNDS=$sr python tools/bench_kernels.py --mode solve --config cfg4 --batch $b > gpurun_out/srm${sr}_solve_cfg4_b$b.log 2>&1
done
done
