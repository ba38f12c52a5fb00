#!/usr/bin/env python3
"""Stand-alone bit reversal: time and rate (2 N sizeof(T) bytes per call) for f64 and f32 at the given log2 sizes -- the
library named by PHASTFT_HIP_LIB (tools/ab.sh-style A/B of generations and thresholds).  Exactness: bitrev_check.py and the tests."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P

for L in [int(a) for a in sys.argv[1:]] or [20, 22, 24, 25, 26, 27, 28]:
    n = 1 << L
    out = []
    for dt, fn, sz in ((torch.float64, P.bit_rev_bravo_f64, 8), (torch.float32, P.bit_rev_bravo_f32, 4)):
        x = torch.arange(n, dtype=dt, device="cuda")
        fn(x, L); fn(x, L)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn(x, L)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        out.append(f"{'f64' if sz == 8 else 'f32'} {ms * 1e3:8.1f} us {2 * n * sz / ms / 1e6:6.0f} GB/s")
        del x
    print(f"2^{L}: " + "   ".join(out), flush=True)
