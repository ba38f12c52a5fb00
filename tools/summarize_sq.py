#!/usr/bin/env python3
"""Per-kernel SQ counter means from a rocprofv3 --pmc run (CSV): where the wave cycles of the pass kernels go.

    python tools/summarize_sq.py <dir> <out.txt> "<command>"
WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles),
MI355X_MICROARCH.md section "rocprofv3 PMC slots".
"""
import csv
import glob
import os
import sys
from collections import OrderedDict, defaultdict

d, out, cmd = sys.argv[1:4]
path = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[-1]
acc = OrderedDict()
with open(path) as f:
    for row in csv.DictReader(f):
        k = row["Kernel_Name"]
        if "phast::" not in k or "fill" in k:
            continue
        acc.setdefault(k, defaultdict(list))[row["Counter_Name"]].append(float(row["Counter_Value"]))
names = []
for c in acc.values():
    for n in c:
        if n not in names:
            names.append(n)
lines = [f"rocprofv3 --kernel-trace --pmc {' '.join(names)} --output-format csv -- {cmd}", ""]
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    lines.append(k)
    lines.append("  " + "  ".join(f"{n}={v:.3g}" for n, v in m.items()))
    if "SQ_WAIT_ANY" in m:
        lines.append(f"  of the wave cycles: parked on waitcnt/barrier {100 * m['SQ_WAIT_ANY'] / wc:.0f} %, issue stall "
                     f"{100 * m.get('SQ_WAIT_INST_ANY', 0) / wc:.0f} %, issuing {100 * m.get('SQ_ACTIVE_INST_ANY', 0) / wc:.0f} %")
    if "SQ_LDS_BANK_CONFLICT" in m and "SQ_WAIT_INST_LDS" in m:
        lines.append(f"  LDS: issue stall on LDS {100 * m['SQ_WAIT_INST_LDS'] / wc:.0f} % of the wave cycles; "
                     f"{m['SQ_LDS_BANK_CONFLICT']:.3g} bank-conflict cycles")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
