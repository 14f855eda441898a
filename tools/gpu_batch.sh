set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export MO_GIT_HEAD=$(cat gpurun_out/.head 2>/dev/null)
timeout -k 10 900 bash tools/profile_all.sh > gpurun_out/r3_profile_all.log 2>&1 || { tail -20 gpurun_out/r3_profile_all.log; exit 1; }
python3 tools/collect_profiles.py r03 | tee gpurun_out/r3_profile_table.txt
du -sh gpurun_out
python bench.py > gpurun_out/r3_bench_b.json 2> gpurun_out/r3_bench_b.err; tail -c 300 gpurun_out/r3_bench_b.json
