#!/usr/bin/env python3
"""The Complex<T> <-> planes sweeps (csrc/complex_nums.hip) against this box's copy rate: every scalar read once, written once.
    python tools/complex_nums_rate.py [log2 of the number of Complex<T> elements, default 27]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P

L = int(sys.argv[1]) if len(sys.argv) > 1 else 27
probe = P.stream_probe(1024, 5)
print(f"# {P.device_info()['name']}: copy probe {probe['copy']:.0f} GB/s (read {probe['read']:.0f}, write {probe['write']:.0f})")
for tdt, name in ((torch.float64, "f64"), (torch.float32, "f32")):
    n = 1 << L
    z = torch.empty(2 * n, dtype=tdt, device="cuda").uniform_(-1, 1)
    esz = z.element_size()
    for label, fn in (("deinterleave", lambda: P.deinterleave(z)), ):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); a, b = fn(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print(f"{label:13s} {name} 2^{L} complex: {1e3 * best:8.1f} us = {2 * 2 * n * esz / best / 1e6:7.0f} GB/s ({2 * 2 * n * esz / best / 1e6 / probe['copy']:.2f} of the copy probe)")
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); w = P.combine_re_im(a, b); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    assert torch.equal(torch.view_as_real(w).reshape(-1), z)
    print(f"{'combine_re_im':13s} {name} 2^{L} complex: {1e3 * best:8.1f} us = {2 * 2 * n * esz / best / 1e6:7.0f} GB/s ({2 * 2 * n * esz / best / 1e6 / probe['copy']:.2f} of the copy probe)")
    del z, a, b, w
    torch.cuda.empty_cache()
