#!/usr/bin/env python3
"""rocprofv3 evidence for one program: a `--kernel-trace --stats` pass, then one `--pmc` pass per counter group (counter passes
carry --kernel-trace only -- never a trace domain -- as the MI355X guide and gpurun require), merged into
gpurun_out/prof_<tag>/summary.json.  The profiled program goes directly behind `--` (python3 <script> ...; no env / bash hop).

usage: python3 tools/profile.py <tag> [--stats-only] [--batch B] -- <script.py> [args...]
This driver itself never touches the GPU.  Summary `_meta`: kernel source digest (bench.py quotes the traffic only while it matches),
git head as passed in MO_GIT_HEAD (the GPU box has no .git), per-kernel effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time of
the SAME pass (MI355X_MICROARCH.md, DVFS give-back)."""
import collections
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PMC_GROUPS = [
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES",
    "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT",
    "FETCH_SIZE",
    "WRITE_SIZE",
    "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_F64 SQ_WAVES GRBM_GUI_ACTIVE",
]


def digest():
    h = hashlib.sha256()
    for name in ("kkt_fused.hip", "kkt_fused_f32.hip", "kkt_generic.hip", "mo_kernels.h"):
        with open(os.path.join(ROOT, "mini_opt_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def main():
    argv = sys.argv[1:]
    sep = argv.index("--")
    opts, prog = argv[:sep], argv[sep + 1:]
    tag = opts[0]
    stats_only = "--stats-only" in opts
    batch = int(opts[opts.index("--batch") + 1]) if "--batch" in opts else None
    out = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    prog = ["python3", os.path.join(ROOT, prog[0])] + prog[1:]

    def run(name, flags):
        d = os.path.join(out, name)
        subprocess.run(["rm", "-rf", d])
        with open(os.path.join(out, name + ".log"), "w") as log:
            rc = subprocess.run(["rocprofv3", "--kernel-trace"] + flags + ["--output-format", "csv", "-d", d, "--"] + prog,
                                cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT).returncode
        print(f"{tag}: pass {name} ({' '.join(flags)}): exit {rc}", flush=True)
        return d

    summary = {"_meta": {"kernel_digest": digest(), "git_head": os.environ.get("MO_GIT_HEAD"), "batch": batch,
                         "command": " ".join(prog[1:]).replace(ROOT + "/", "")}}
    d = run("stats", ["--stats"])
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "kkt_" in row["Name"] or "nls_" in row["Name"]:
                summary.setdefault(row["Name"][:60], {})["kernel_stats"] = {k: row[k] for k in ("Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")}
        subprocess.run(["cp", f, os.path.join(out, "kernel_stats.csv")])
    try:
        summary["_meta"]["program_output"] = [ln for ln in open(os.path.join(out, "stats.log")).read().splitlines() if ln.startswith("{")][-1]
    except Exception:
        pass
    if not stats_only:
        for i, grp in enumerate(PMC_GROUPS, 1):
            d = run(f"pmc{i}", ["--pmc"] + grp.split())
            dur = collections.defaultdict(list)  # kernel time inside THIS pass (for the clock estimate)
            for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    dur[row["Kernel_Name"][:60]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                agg = collections.defaultdict(lambda: collections.defaultdict(float))
                cnt = collections.defaultdict(int)
                for row in csv.DictReader(open(f)):
                    k = row["Kernel_Name"][:60]
                    if "kkt_" not in k and "nls_" not in k:
                        continue
                    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                    cnt[(k, row["Counter_Name"])] += 1
                for k, cs in agg.items():
                    for c, v in cs.items():
                        summary.setdefault(k, {})[c] = v / cnt[(k, c)]
                    if "GRBM_GUI_ACTIVE" in cs and dur.get(k):
                        ns = sum(dur[k]) / len(dur[k])
                        summary[k]["counter_pass_kernel_ns"] = ns
                        summary[k]["effective_clock_ghz"] = summary[k]["GRBM_GUI_ACTIVE"] / 8.0 / ns
    json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
    if "--keep-raw" not in opts:   # the raw per-dispatch tables are 20-40 MB per tag; gpurun merges at most 64 MiB back
        for name in ["stats"] + [f"pmc{i}" for i in range(1, len(PMC_GROUPS) + 1)]:
            subprocess.run(["rm", "-rf", os.path.join(out, name)])
    for k, v in summary.items():
        if k != "_meta":
            print(k, json.dumps({c: v[c] for c in ("kernel_stats", "FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "effective_clock_ghz") if c in v}))


if __name__ == "__main__":
    main()
