set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "beyond_every or large_fp64 or generic_solve_first" > gpurun_out/r3_tests_b.log 2>&1
rc=$?; echo "large tests rc=$rc"; tail -30 gpurun_out/r3_tests_b.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 700 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r3_tests_b_all.log 2>&1
rc=$?; echo "all tests rc=$rc"; tail -15 gpurun_out/r3_tests_b_all.log
