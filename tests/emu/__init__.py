"""CPU emulator of the tile kernels (TEST INFRASTRUCTURE: never imported by the product package).

`emu.hip` includes the product's kernel headers (phastft_amd/csrc/tile_fft.hpp, row_fft.hpp, plan.hpp) and runs
their `__host__ __device__` phase functions thread by thread on the host, so index arithmetic, LDS layouts,
twiddle tables and plans are checked against the oracle in the GPU-less build container (tests/test_emulator.py).
"""
from __future__ import annotations

import os
import subprocess

from phastft_amd import build as _b

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_LIB = os.path.join(HERE, "libphastft_emu.so")


def build_emulator(force: bool = False) -> str:
    src = os.path.join(HERE, "emu.hip")
    if force or _b._stale(EMU_LIB, [src] + _b._deps()):
        # host code only: the kernels in the headers are compiled for the device but never launched
        cmd = [_b.hipcc(), *_b.FLAGS, "-I", _b.INCLUDE, "-I", _b.SRC, "-shared", src, "-o", EMU_LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for emu:\n{r.stdout}\n{r.stderr}")
    return EMU_LIB
