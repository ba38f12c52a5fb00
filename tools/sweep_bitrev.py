#!/usr/bin/env python3
"""Bit-reversal kernel variants (PHAST_BITREV_VARIANT) at 2^20..2^28; roofline = 2*N*sizeof(T) / t."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import phastft_amd as P  # noqa: E402

for dt, fn, sz in ((torch.float64, P.bit_rev_bravo_f64, 8), (torch.float32, P.bit_rev_bravo_f32, 4)):
    for log_n in (20, 24, 26, 28, 30) if len(sys.argv) < 2 else [int(a) for a in sys.argv[1:]]:
        n = 1 << log_n
        x = torch.arange(n, dtype=dt, device="cuda")
        for variant in (0, 5, 6, 7, 8):
            os.environ["PHAST_BITREV_VARIANT"] = str(variant)
            fn(x, log_n)
            fn(x, log_n)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record()
            for _ in range(reps):
                fn(x, log_n)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            ok = bool((x == torch.arange(n, dtype=dt, device="cuda")).all())  # even number of applications
            print(f"{str(dt)[6:]} 2^{log_n} variant {variant}: {ms * 1e3:9.1f} us  {2 * n * sz / ms / 1e6:7.0f} GB/s  involution_ok={ok}", flush=True)
        del x
