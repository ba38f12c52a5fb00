#!/usr/bin/env python3
"""Generates tests/golden/fft_golden.npz: inputs and expected outputs for the FFT path.

The reference (Rust) cannot be imported or built in the build container and ships no data files, so
these vectors come from an INDEPENDENT FFT -- numpy's pocketfft evaluated in long double -- exactly the
role RustFFT plays in the reference's own tests (lib.rs:298-338).  Inputs follow the reference's test
inputs: the ramp re = im = 1..N (lib.rs:310-311, r2c.rs:918) and seeded uniform [-1, 1) signals
(utilities/src/lib.rs:26-75).  Expected outputs are stored in f64.

    python tests/golden/make_golden.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def fft_ld(z):
    return np.fft.fft(z.astype(np.clongdouble))


def main():
    out = {}
    rng = np.random.default_rng(0xCAFE)
    for k in (4, 6, 8, 10, 12):
        n = 1 << k
        ramp = np.arange(1, n + 1, dtype=np.float64)
        for dt in ("f64", "f32"):
            dtype = np.float64 if dt == "f64" else np.float32
            for kind, (re, im) in (("ramp", (ramp, ramp)), ("rand", (rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)))):
                re, im = re.astype(dtype), im.astype(dtype)  # the input the implementation actually sees
                spec = fft_ld(re.astype(np.longdouble) + 1j * im.astype(np.longdouble))
                tag = f"{kind}_{dt}_{k}"
                out["in_" + tag] = np.stack([re, im]).astype(np.float64)
                out["re_" + tag] = np.asarray(spec.real, dtype=np.float64)
                out["im_" + tag] = np.asarray(spec.imag, dtype=np.float64)
            x = rng.uniform(-1, 1, n).astype(dtype)
            spec = fft_ld(x.astype(np.longdouble) + 0j)[: n // 2 + 1]
            tag = f"r2c_{dt}_{k}"
            out["in_" + tag] = x.astype(np.float64)
            out["re_" + tag] = np.asarray(spec.real, dtype=np.float64)
            out["im_" + tag] = np.asarray(spec.imag, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "fft_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "fft_golden.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
