#!/usr/bin/env python3
"""python tools/bitrev_check.py [L ...]: the stand-alone f64 bit reversal of 2^L points (and of a
batch of four) against the permutation computed with integer tensor ops; then its time and rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from phastft_amd import _lib
lib = _lib.lib()

def rev_index(L, device):
    i = torch.arange(1 << L, dtype=torch.int64, device=device)
    r = torch.zeros_like(i)
    for b in range(L):
        r |= ((i >> b) & 1) << (L - 1 - b)
    return r

for L in [int(a) for a in sys.argv[1:]] or [14, 15, 20, 21, 26]:
    for batch in (1, 4) if L <= 22 else (1,):
        n = 1 << L
        x = torch.arange(n * batch, dtype=torch.float64, device="cuda")
        rc = lib.phast_bit_rev_f64_dev(C.c_void_p(x.data_ptr()), C.c_uint(L), C.c_size_t(batch), C.c_size_t(n), C.c_void_p(0))
        assert rc == 0, rc
        torch.cuda.synchronize()
        want = rev_index(L, "cuda").to(torch.float64)
        for b in range(batch):
            assert torch.equal(x[b * n:(b + 1) * n], want + b * n), (L, batch, b)
    x = torch.arange(n, dtype=torch.float64, device="cuda")
    run = lambda: lib.phast_bit_rev_f64_dev(C.c_void_p(x.data_ptr()), C.c_uint(L), C.c_size_t(1), C.c_size_t(n), C.c_void_p(0))
    run(); run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"2^{L}: exact; {ms * 1e3:8.1f} us  {2 * n * 8 / ms / 1e6:6.0f} GB/s", flush=True)
