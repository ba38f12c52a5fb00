#!/usr/bin/env python3
"""Where does the stand-alone bit reversal lose its bandwidth?  Same bytes (512 MiB of f64), different strides:
one array of 2^26 (rows of a tile 8 MiB apart) against batches of smaller arrays (rows 2^(L-6) elements apart)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import phastft_amd as P
from phastft_amd import _lib
lib = _lib.lib()
total = 26
for variant in (0, 5, 7):
    os.environ["PHAST_BITREV_VARIANT"] = str(variant)
    for L in (26, 24, 22, 20, 18, 16, 14, 12):
        batch = 1 << (total - L)
        x = torch.arange(1 << total, dtype=torch.float64, device="cuda")
        def run():
            rc = lib.phast_bit_rev_f64_dev(C.c_void_p(x.data_ptr()), C.c_uint(L), C.c_size_t(batch), C.c_size_t(1 << L), C.c_void_p(0))
            assert rc == 0
        run(); run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"variant {variant}: {batch:6d} x 2^{L}: {ms*1e3:8.1f} us  {2 * (1 << total) * 8 / ms / 1e6:6.0f} GB/s", flush=True)
