set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1
for i in 1 2; do
python tools/bench_kernels.py --mode step --shape 96,10,48,192 --batch 32768 > gpurun_out/n96_step_la$i.log 2>&1
MO_LIB_PATH=tools/ab_libs/libminiopt_la8.so python tools/bench_kernels.py --mode step --shape 96,10,48,192 --batch 32768 > gpurun_out/n96_step_nola$i.log 2>&1
done
python tools/bench_kernels.py --mode solve --shape 96,10,48,192 --batch 32768 > gpurun_out/n96_solve_la.log 2>&1
MO_LIB_PATH=tools/ab_libs/libminiopt_la8.so python tools/bench_kernels.py --mode solve --shape 96,10,48,192 --batch 32768 > gpurun_out/n96_solve_nola.log 2>&1
