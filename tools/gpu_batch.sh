set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
MO_FUZZ_EXTRA_SEEDS=100-139 timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider -s > gpurun_out/r3_fuzz_soak.log 2>&1
rc=$?; echo "soak rc=$rc"; grep -c "differ from the oracle" gpurun_out/r3_fuzz_soak.log; grep "differ from the oracle" gpurun_out/r3_fuzz_soak.log | grep -v " 0 of" | head -40; tail -5 gpurun_out/r3_fuzz_soak.log
