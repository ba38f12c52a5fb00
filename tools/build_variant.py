#!/usr/bin/env python3
"""A library VARIANT for an A/B (tools/ab_single.py --libs): only the named translation units are recompiled with the extra
flags, every other object is the product's -- a variant of one kernel costs one hipcc run, not a rebuild of the library.

    python tools/build_variant.py --tag _x2 --units wave_f64,wave_f32 -- -DPHAST_WAVE_TILES_PER_BLOCK=2

writes phastft_amd/lib/libphastft_hip<tag>.so (git-ignored, travels to the GPU box with the snapshot)."""
import argparse, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phastft_amd import build as B

ap = argparse.ArgumentParser()
ap.add_argument("--tag", required=True)
ap.add_argument("--units", required=True)
ap.add_argument("flags", nargs="*")
a = ap.parse_args()
B.build()  # the product objects, up to date
units = a.units.split(",")
objs = []
for u in B.UNITS:
    if u in units:
        obj = os.path.join(B.OBJ, u + a.tag + ".o")
        cmd = [B.hipcc(), *[f for f in B.FLAGS if not f.startswith("-Rpass")], *B.UNIT_FLAGS.get(u, []), *a.flags, "-I", B.INCLUDE, "-c",
               os.path.join(B.SRC, u + ".hip"), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-3000:])
        objs.append(obj)
    else:
        objs.append(os.path.join(B.OBJ, u + ".o"))
lib = B.LIB.replace(".so", a.tag + ".so")
r = subprocess.run([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr[-3000:])
print(lib)
