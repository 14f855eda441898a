for shp in "160,16,32,170 2048" "200,20,64,256 2048" "256,40,128,300 2048" "384,32,96,400 1024" "512,40,128,600 512"; do
  set -- $shp
  echo -n "$1 @ $2: "
  timeout -k 10 200 python tools/bench_kernels.py --mode step --shape $1 --batch $2 --reps 3 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step %.3f ms = %.0f steps/s' % (d['ms_mean'], d['units_per_s']), end='; ')"
  timeout -k 10 200 python tools/bench_kernels.py --mode solve --shape $1 --batch $2 --reps 2 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('Solve %.1f ms = %.0f solves/s at %.1f iterations' % (d['ms_mean'], d['units_per_s'], d['mean_iterations']))"
done
