#!/usr/bin/env python3
"""Sweep the wave-tile stagger (wave_fft.hpp: PHAST_WAVE_STAGGER="units,mask") on the single-transform plans: one bench.py
headline run per setting (fresh process: the setting is read once per process)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
settings = sys.argv[1:] or ["0,0", "4,3", "6,3", "8,3", "10,3", "12,3", "4,7", "6,7", "8,7", "8,1", "16,1"]
for s in settings:
    env = dict(os.environ, PHAST_WAVE_STAGGER=s)
    vals = []
    for rep in range(2):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-configs", "--no-scaling-reference"],
                             env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        vals.append((d["ms_per_step"] * 1e3, [round(x * 1e3, 2) for x in d["roofline"]["pass_ms"]]))
    print(f"stagger {s:>6}: " + "   ".join(f"{v[0]:6.2f} us {v[1]}" for v in vals), flush=True)
