#!/usr/bin/env python3
"""Turn rocprofv3 CSV output into the summaries committed under profiles/.

    python tools/summarize_prof.py stats  <dir> <out.csv>            # copy the --stats kernel table
    python tools/summarize_prof.py pmc    <fetch_dir> <write_dir> <out.txt> <out.json> <key> "<command>"

PMC: FETCH_SIZE and WRITE_SIZE are collected in separate passes (MI355X_MICROARCH.md, rocprofv3 PMC slots), both
in KiB per dispatch; HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 tallies 128-byte read
requests at 64 B).  The JSON records, per tile_fft kernel in launch order of one transform, the mean bytes per
launch; bench.py reads it for roofline.traffic.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import OrderedDict


def find(d, pat):
    hits = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    if not hits:
        raise SystemExit(f"no {pat} under {d}")
    return hits[-1]


def counter(d, name):
    per = OrderedDict()
    with open(find(d, "*counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            kn = row.get("Kernel_Name", "")
            if row.get("Counter_Name") != name or not any(t in kn for t in ("tile_fft_kernel", "wave_fft_kernel", "quad_fft_kernel", "untangle_kernel", "r2c_last_pass_kernel", "c2r_first_pass_kernel", "c2r_preprocess_kernel")):
                continue
            per.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return per


if sys.argv[1] == "stats":
    shutil.copy(find(sys.argv[2], "*kernel_stats.csv"), sys.argv[3])
    print(open(sys.argv[3]).read())
else:
    fetch_dir, write_dir, out_txt, out_json, key, cmd = sys.argv[2:8]
    alg = int(sys.argv[8]) if len(sys.argv) > 8 else 33554432
    fe, wr = counter(fetch_dir, "FETCH_SIZE"), counter(write_dir, "WRITE_SIZE")
    lines = [f"rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- {cmd}",
             f"rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -- {cmd}",
             "(separate passes, as MI355X_MICROARCH.md 'rocprofv3 PMC slots' requires; counters are KiB per dispatch)", "",
             "kernel, counter, dispatches, mean, min, max"]
    for nm, per in (("FETCH_SIZE", fe), ("WRITE_SIZE", wr)):
        for k, v in per.items():
            lines.append(f"{k}, {nm}, {len(v)}, {sum(v) / len(v):.1f}, {min(v):.1f}, {max(v):.1f}")
    lines += ["", "HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   [FETCH_SIZE doubled: gfx950 tallies "
              "128-B requests at 64 B, MI355X_MICROARCH.md section HBM]"]
    kernels = []
    for k in fe:
        if k not in wr:
            continue
        b = (2 * sum(fe[k]) / len(fe[k]) + sum(wr[k]) / len(wr[k])) * 1024
        lines.append(f"{k}: {b:.0f} bytes  (algorithmic {alg}; ratio {b / alg:.4f})")
        kernels.append({"kernel": k, "hbm_bytes_per_launch": b})
    open(out_txt, "w").write("\n".join(lines) + "\n")
    try:
        t = json.load(open(out_json))
    except (OSError, ValueError):
        t = {}
    t[key] = {"kernels": kernels,
              "source": f"{os.path.basename(out_txt)}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on "
                        f"`{cmd}`, (2*FETCH_SIZE + WRITE_SIZE)*1024"}
    json.dump(t, open(out_json, "w"), indent=1)
    print("\n".join(lines))
