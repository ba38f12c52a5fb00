"""Builds phastft_amd/lib/libphastft_hip.so from phastft_amd/csrc/*.hip with hipcc for gfx950.

    python -m phastft_amd.build [--force] [--jobs N]

The library is the product: hand-written HIP kernels + the C ABI of include/phastft_hip.h.  hipcc
cross-compiles without a GPU, so this runs in the build container and the .so travels to the GPU box.
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import json
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "build")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libphastft_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

UNITS = ["c_abi", "tile_f64_a", "tile_f64_bc", "tile_f64_bc_wide", "tile_f32_a", "tile_f32_bc", "tile_f32_bc_wide", "tile_f64_r2c", "tile_f32_r2c", "tile_f64_c2r", "tile_f32_c2r", "wave_f64", "wave_f32", "quad_f64", "quad_f32", "small_fft", "bitrev", "complex_nums", "r2c", "fill", "probe", "twiddle"]
# built only with --experimental (lib/libphastft_hip_exp.so): nothing at present (round 6: the f32 wave tiles were rebuilt on
# float2 column pairs and moved into the product)
EXPERIMENTAL_UNITS = []
EXPERIMENTAL_FLAGS = ()
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-ffp-contract=fast", "-Rpass-analysis=kernel-resource-usage"]
# per-unit compiler options (the scheduling strategy is a translation-unit option: tile_dispatch.hpp says why)
UNIT_FLAGS = {"tile_f64_bc_wide": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
              "tile_f32_bc_wide": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp"],
              # the f32 wave / four-wave tiles run one wave per SIMD: registers are free, latency is not -- max-ilp issues the 15
              # step-twiddle LDS reads of a tile up front instead of two ahead of their use (one f32 transform of 2^20 points
              # 19.45 -> 19.10 us; the f64 twins did not move: profiles/r06_max_ilp_ab.log)
              "wave_f32": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
              "quad_f32": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
RESOURCES = os.path.join(LIB_DIR, "kernel_resources.json")  # per kernel: VGPRs, scratch bytes per lane, occupancy, spills


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X library cannot be built without ROCm")
    return exe


def _deps(src: str | None = None) -> list[str]:
    """What a unit's object depends on: the headers it includes (followed recursively through csrc/ and include/), the
    public header and this file -- a change to the host side (planner*.hpp ...) does not recompile the kernel units."""
    fixed = [os.path.join(INCLUDE, "phastft_hip.h"), os.path.abspath(__file__)]
    if src is None:
        return [os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith((".hpp", ".h"))] + fixed
    seen, todo = set(), [src]
    while todo:
        f = todo.pop()
        try:
            text = open(f).read()
        except OSError:
            continue
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, flags=re.M):
            for base in (os.path.dirname(f), SRC, INCLUDE):
                cand = os.path.normpath(os.path.join(base, inc))
                if os.path.exists(cand):
                    if cand not in seen:
                        seen.add(cand)
                        todo.append(cand)
                    break
    return sorted(seen) + fixed


def _stale(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _compile(unit: str, force: bool, trace: bool = False, extra: tuple = (), tag: str = "") -> str:
    src = os.path.join(SRC, unit + ".hip")
    obj = os.path.join(OBJ, unit + ("_trace" if trace else "") + tag + ".o")
    if force or _stale(obj, [src] + _deps(src)):
        cmd = [hipcc(), *FLAGS, *(["-DPHAST_TRACE"] if trace else []), *UNIT_FLAGS.get(unit, []), *extra, "-I", INCLUDE, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {unit}:\n{r.stdout}\n{r.stderr}")
        with open(obj + ".res.json", "w") as f:
            json.dump(_resource_usage(r.stderr), f)
    return obj


def _resource_usage(remarks: str) -> dict:
    """{mangled kernel name: {"vgprs", "agprs", "scratch", "occupancy", "vgpr_spill", "sgpr_spill"}} out of the compiler's
    kernel-resource-usage remarks: a kernel that starts to use scratch memory (a register array the optimiser could
    not keep in registers, or spills) pays for it in HBM traffic -- tests/test_kernel_resources.py watches this."""
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
            "VGPRs Spill": "vgpr_spill", "SGPRs Spill": "sgpr_spill"}
    out, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"remark:\s+(.*?):\s+(\S+)\s+\[-Rpass-analysis", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = out.setdefault(v, {})
        elif cur is not None and k in keys:
            cur[keys[k]] = int(v)
    return out


def build(force: bool = False, jobs: int | None = None, verbose: bool = False, trace: bool = False,
          extra: tuple = (), tag: str = "", experimental: bool = False) -> str:
    """trace=True builds lib/libphastft_hip_trace.so with per-phase s_memtime stamps (tools/trace_tile.py;
    load it with PHASTFT_HIP_LIB=...); the product library never carries them."""
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    units = list(UNITS)
    if experimental:
        units += EXPERIMENTAL_UNITS
        extra = tuple(extra) + EXPERIMENTAL_FLAGS
        tag = tag + "_exp"
    jobs = jobs or min(len(units), os.cpu_count() or 4)
    lib = LIB.replace(".so", ("_trace" if trace else "") + tag + ".so")  # tag/extra: experimental variants (tools/)
    with cf.ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(lambda u: _compile(u, force, trace, extra, tag), units))
    if force or _stale(lib, objs):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if not trace and not tag:
        merged = {}
        for o in objs:
            if os.path.exists(o + ".res.json"):
                merged.update(json.load(open(o + ".res.json")))
        with open(RESOURCES, "w") as f:
            json.dump(merged, f, indent=1, sort_keys=True)
    if verbose:
        print(lib)
    return lib


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--experimental", action="store_true", help="lib/libphastft_hip_exp.so: + the f32 wave tiles")
    a = ap.parse_args()
    build(a.force, a.jobs, verbose=True, trace=a.trace, experimental=a.experimental)
    sys.exit(0)
