#!/usr/bin/env python3
"""Static instruction histogram of one kernel in a hipcc -S listing: tools/isa_hist.py listing.s 'kkt_fused_f64_kernel<8, 1, 3, false, 1, 0, 1>'
(the fused kernels are almost straight-line code, so the static mix is close to the dynamic one outside the J loop)."""
import collections
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2]
for p in re.split(r"\n\t\.globl\t", txt)[1:]:
    name = p.split(None, 1)[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if want not in dem:
        continue
    body = p.split(".end_amdhsa_kernel")[0]
    c = collections.Counter()
    for line in body.split("\n"):
        m = re.match(r"\t([a-z_0-9]+)", line)
        if m and not line.startswith("\t."):
            c[m.group(1)] += 1
    groups = collections.Counter()
    for k, v in c.items():
        if k.startswith("v_mfma"): groups["mfma"] += v
        elif k.startswith("v_accvgpr"): groups["accvgpr moves"] += v
        elif k.startswith("v_"): groups["other valu"] += v
        elif k.startswith("s_"): groups["salu/branch/wait"] += v
        elif k.startswith("ds_"): groups["lds"] += v
        elif k.startswith("scratch_"): groups["scratch"] += v
        elif k.startswith(("global_", "buffer_", "flat_")): groups["vmem"] += v
    print(dem[:100])
    print("  ", dict(groups))
    for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 25):
        print("   %-28s %d" % (k, v))
