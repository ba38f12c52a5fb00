#!/usr/bin/env python3
"""How much of each parity gate the HIP path uses: summarises the file `PHAST_RECORD_ERRORS=<path> python -m pytest tests -m gpu`
wrote (tests/tolerances.py: one JSON line per comparison) into profiles/r06_gate_usage.txt.

    python tools/gate_usage.py gpurun_out/r06_errors.jsonl > profiles/r06_gate_usage.txt
"""
import collections
import json
import re
import sys

rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
by = collections.defaultdict(list)
for r in rows:
    tag = re.split(r"[ =]", r["tag"])[0]          # the test's label without its parameters
    by[tag].append(r)


def worst(rs, key, gate):
    best = (0.0, None)
    for r in rs:
        if r[gate] > 0 and r[key] / r[gate] >= best[0]:
            best = (r[key] / r[gate], r)
    return best


out = []
for tag, rs in by.items():
    (ur, rr), (ub, rb) = worst(rs, "rel", "gate_rel"), worst(rs, "bin", "gate_bin")
    out.append((ur, tag, len(rs), rr, ub, rb))
out.sort(key=lambda t: -t[0])
overall_rel = max(t[0] for t in out)
overall_bin = max(t[4] for t in out)
med = sorted(t[0] for t in out)[len(out) // 2]
print("# round 6: how much of each parity gate the HIP path uses -- every comparison of `python -m pytest tests -m gpu` that goes through tests/tolerances.py")
print(f"# (PHAST_RECORD_ERRORS; {len(rows)} comparisons in one run of the suite on an MI355X; tools/gate_usage.py).  Per test tag: comparisons, the WORST measured / gate")
print("# ratio for the rel-L2 gate and for the worst-bin gate, with (log2 N, measured, gate) of that worst case.  A ratio of 1 would be a failure.")
print(f"# overall worst: rel-L2 {overall_rel:.3f}, worst bin {overall_bin:.3f} of the gate; median tag {med:.3f}")
for ur, tag, n, rr, ub, rb in out:
    a = f"({rr['log2n']}, {rr['rel']}, {rr['gate_rel']})" if rr else "-"
    b = f"({rb['log2n']}, {rb['bin']}, {rb['gate_bin']})" if rb else "-"
    print(f"{tag:<44} n={n:5d}  rel {ur:.3f} {a}   bin {ub:.3f} {b}")
