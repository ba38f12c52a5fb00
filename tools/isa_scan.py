#!/usr/bin/env python3
"""CPU-only: compile every translation unit of the library to gfx950 assembly and look for SERIALISED LOADS -- a global load
followed by `s_waitcnt vmcnt(0)` before the next load goes out, several times in a row: each one is a full memory (or L2)
round trip the wave sits through alone.  Round 4 found the one-pass kernel's Complex<T> pair loads (16 in a row, fixed:
2^12 on pairs 11.7 -> 8.6 us) and the fused R2C untangle's table loads (16 in a row, now 4 x 4) this way.  A chain is not
always a bug: the compiler serialises on purpose where keeping the loads in flight would cost registers past an occupancy
step (the 32-point f32 first passes, the f64 fused C2R first pass: profiles/HISTORY.md section 9a) -- compare
phastft_amd/lib/kernel_resources.json before and after a change.
    python tools/isa_scan.py [--min 3] [--show KERNEL_SUBSTRING]"""
import argparse, concurrent.futures as cf, os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phastft_amd import build as B

ap = argparse.ArgumentParser()
ap.add_argument("--min", type=int, default=3, help="report kernels with at least this many (load, full wait) groups in a row")
ap.add_argument("--show", default="", help="print the load / wait / barrier / store sequence of kernels whose mangled name contains this")
a = ap.parse_args()
out_dir = tempfile.mkdtemp(prefix="phast_isa_")


def compile_unit(u):
    out = os.path.join(out_dir, u + ".s")
    flags = [f for f in B.FLAGS if not f.startswith("-Rpass") and f != "-fPIC"]
    cmd = [B.hipcc(), *flags, *B.UNIT_FLAGS.get(u, []), "-I", B.INCLUDE, "--cuda-device-only", "-S", "-o", out, os.path.join(B.SRC, u + ".hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr[-2000:])
    return out


with cf.ThreadPoolExecutor(8) as ex:
    files = list(ex.map(compile_unit, B.UNITS))
rows, total = [], 0
for fn in files:
    for f in re.split(r"\n(?=_Z\w+:)", open(fn).read()):
        m = re.match(r"(_Z\w+):", f)
        if not m:
            continue
        total += 1
        seq = []
        for line in f.split(".Lfunc_end")[0].splitlines():
            t = line.strip()
            if t.startswith(("global_load", "buffer_load", "flat_load")):
                seq.append("L")
            elif t.startswith("s_waitcnt") and "vmcnt" in t:
                seq.append("W" + re.search(r"vmcnt\((\d+)\)", t).group(1))
            elif t.startswith("global_store"):
                seq.append("S")
            elif t.startswith("s_barrier"):
                seq.append("|")
        compact = "".join("W" if s == "W0" else "L" if s == "L" else "." for s in seq)
        runs = re.findall(r"(?:L{1,2}W){%d,}" % a.min, compact)
        best = max((r.count("W") for r in runs), default=0)
        if best:
            rows.append((best, os.path.basename(fn), m.group(1)))
        if a.show and a.show in m.group(1):
            s = re.sub(r"(?:S ){4,}", "S* ", " ".join(seq))
            print(m.group(1), "\n   ", s, "\n")
rows.sort(reverse=True)
print(f"{total} kernels; {len(rows)} with >= {a.min} (one or two loads, full wait) groups in a row:")
for best, fn, name in rows:
    print(f"  {best:3d}  {fn:22s} {name}")
