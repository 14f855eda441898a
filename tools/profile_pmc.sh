#!/bin/bash
# Collect rocprofv3 kernel-trace stats and PMC counters (separate passes, as the MI355X guide prescribes) for bench.py.
# usage: tools/profile_pmc.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline $*"
# pass 0: the default bench.py command (50 timed + 10 warm-up launches), so that the --stats average is comparable with the
# event-timed kernel_ms of the bench line; the counter passes below use a short run
if [ -z "${SKIP_STATS:-}" ]; then
rm -rf $OUT/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --no-cpu-baseline $* > $OUT/stats.log 2>&1
fi
if [ -n "${STATS_ONLY:-}" ]; then tail -1 $OUT/stats.log; exit 0; fi
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_F64 SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 $ROOT/bench.py $ARGS > $OUT/pmc$i.log 2>&1
  echo "pass $i ($PMC): exit $?"
done
python3 - <<PY
import csv, glob, collections, os, json
out = "$OUT"
summary = {}
for d in sorted(glob.glob(out + "/pmc*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "kkt_" not in k: continue
            agg[k[:60]][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[(k[:60], row["Counter_Name"])] += 1
        for k, cs in agg.items():
            for c, v in cs.items():
                print(f"{os.path.basename(os.path.dirname(d))} {k} {c} per-dispatch {v / cnt[(k, c)]:.6g}")
                summary.setdefault(k, {})[c] = v / cnt[(k, c)]
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "kkt_" in row["Name"]:
            summary.setdefault(row["Name"][:60], {})["kernel_stats"] = {k: row[k] for k in ("Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")}
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
PY
