"""CPU emulator of the tile kernels (TEST INFRASTRUCTURE: never imported by the product package).

`emu.hip` includes the product's kernel headers (phastft_amd/csrc/tile_fft.hpp, row_fft.hpp, plan.hpp) and runs
their `__host__ __device__` phase functions thread by thread on the host, so index arithmetic, LDS layouts,
twiddle tables and plans are checked against the oracle in the GPU-less build container (tests/test_emulator.py).
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess

from phastft_amd import build as _b

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_LIB = os.path.join(HERE, "libphastft_emu.so")


def build_emulator(force: bool = False) -> str:
    src = os.path.join(HERE, "emu.hip")
    if force or _b._stale(EMU_LIB, [src] + _b._deps()):
        # host code only (--cuda-host-only: the kernels are never launched), in four parts compiled in parallel at -O1:
        # the template instantiations of every tile shape make one -O3 translation unit a four-minute compile
        flags = [f for f in _b.FLAGS if f != "-O3"] + ["-O1", "--cuda-host-only"]
        os.makedirs(os.path.join(HERE, "build"), exist_ok=True)

        def part(k: int) -> str:
            obj = os.path.join(HERE, "build", f"emu_part{k}.o")
            cmd = [_b.hipcc(), *flags, f"-DEMU_PART={k}", "-I", _b.INCLUDE, "-I", _b.SRC, "-c", src, "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for emu part {k}:\n{r.stdout}\n{r.stderr}")
            return obj

        with cf.ThreadPoolExecutor(4) as ex:
            objs = list(ex.map(part, (1, 2, 3, 4)))
        r = subprocess.run([_b.hipcc(), "-shared", "-fPIC", "-o", EMU_LIB, *objs], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed for emu:\n{r.stdout}\n{r.stderr}")
    return EMU_LIB
