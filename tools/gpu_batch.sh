set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r3_tests_m.log 2>&1
rc=$?; echo "all tests rc=$rc"; tail -30 gpurun_out/r3_tests_m.log
