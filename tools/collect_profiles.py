#!/usr/bin/env python3
"""Copy gpurun_out/prof_<tag>/{summary.json, kernel_stats.csv} of every tag of the round (default r03) into profiles/ and print one line per kernel
(average launch, units/s, instruction mix per problem, HBM bytes) -- the numbers profiles/README.md quotes."""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r03"
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{ROUND}_*"))):
    tag = os.path.basename(d)[5:]
    if not os.path.exists(os.path.join(d, "summary.json")):
        continue
    s = json.load(open(os.path.join(d, "summary.json")))
    shutil.copy(os.path.join(d, "summary.json"), os.path.join(ROOT, "profiles", tag + "_pmc_summary.json"))
    shutil.copy(os.path.join(d, "kernel_stats.csv"), os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
    B = s["_meta"]["batch"]
    for k, v in s.items():
        if k == "_meta" or "kkt_" not in k:
            continue
        avg = float(v["kernel_stats"]["AverageNs"]) / 1e6
        g = lambda n: v.get(n, 0)
        print(f"{tag:26s} {k[32:76]:44s} {avg:8.3f} ms {B / avg / 1e3:7.2f} M/s  MFMA {g('SQ_INSTS_MFMA') / B:6.0f}  other VALU {(g('SQ_INSTS_VALU') - g('SQ_INSTS_MFMA')) / B:7.0f}"
              f"  fetch x2 {g('FETCH_SIZE') * 2048 / 1e9:7.3f} GB  write {g('WRITE_SIZE') * 1024 / 1e9:6.3f} GB  {v.get('effective_clock_ghz', 0):.2f} GHz  digest {s['_meta']['kernel_digest']}")
