"""Element-aligned pointers and odd distances through every `_dev` entry point.

The reference takes plain slices (`&mut [T]`, algorithms/dit.rs:276-300: no alignment contract): a Rust caller is entitled
to pass `&mut v[1..]`, whose pointer is aligned to the ELEMENT only -- 8 bytes for f64, 4 for f32 -- and batches whose
transforms sit an odd number of elements apart.  The kernels use 8- and 16-byte accesses wherever the layout lets them
(pair loads of the real transforms, `Complex<T>` pairs, the 16-byte branch of the stand-alone bit reversal, bitrev.hip),
so every entry point is driven here through `buf[off:]` views of device tensors, off = 1 and 3, at 2^10 (the one-pass
kernel), 2^14 and 2^20 (multi-pass plans), against the oracle / an independent float64 FFT with the gates of
tests/tolerances.py; the elements before the view and behind its end must come back untouched.
"""
import numpy as np
import pytest

from tests import tolerances as tol

pytestmark = pytest.mark.gpu

SIZES = [10, 14, 20]
OFFS = [1, 3]
GUARD = 7.0  # what the elements around the views hold


def _views(torch, tdt, off, *lens):
    """one device tensor per requested length, each a view `buf[off:off + len]` of a larger buffer filled with GUARD"""
    bufs = [torch.full((off + n + 5,), GUARD, dtype=tdt, device="cuda") for n in lens]
    return bufs, [b[off:off + n] for b, n in zip(bufs, lens)]


def _guards_intact(bufs, off, *lens):
    for b, n in zip(bufs, lens):
        h = b.cpu().numpy()
        assert np.all(h[:off] == GUARD) and np.all(h[off + n:] == GUARD), "wrote outside the slice"


@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("k", SIZES)
@pytest.mark.parametrize("off", OFFS)
def test_c2c_planar_on_element_aligned_views(gpu, oracle, dt, k, off):
    """phast_fft_{64,32}_dit_dev, forward and inverse, one transform and a batch an ODD distance apart"""
    import torch

    n = 1 << k
    ndt, tdt = (np.float64, torch.float64) if dt == "f64" else (np.float32, torch.float32)
    planner = (gpu.PlannerDit64 if dt == "f64" else gpu.PlannerDit32)(n)
    fft = gpu.fft_64_dit_with_planner if dt == "f64" else gpu.fft_32_dit_with_planner
    ofn = oracle.fft_64_dit if dt == "f64" else oracle.fft_32_dit
    h_re, h_im = oracle.fill(n, ndt, seed=0xA11, transform_id=k)
    bufs, (re, im) = _views(torch, tdt, off, n, n)
    assert re.data_ptr() % (16 if dt == "f64" else 8) != 0        # really element-aligned only
    re.copy_(torch.from_numpy(h_re)); im.copy_(torch.from_numpy(h_im))
    fft(re, im, gpu.Direction.Forward, planner)
    z = np.fft.fft(h_re.astype(np.float64) + 1j * h_im.astype(np.float64))
    tol.check(f"unaligned_c2c off={off}", dt, k, re.cpu().numpy(), im.cpu().numpy(), z.real, z.imag)
    w_re, w_im = h_re.copy(), h_im.copy()
    ofn(w_re, w_im, oracle.FORWARD)
    tol.check(f"unaligned_c2c_vs_oracle off={off}", dt, k, re.cpu().numpy(), im.cpu().numpy(), w_re.astype(np.float64),
              w_im.astype(np.float64), against="oracle")
    fft(re, im, gpu.Direction.Reverse, planner)
    lim = tol.ROUNDTRIP_ABS[dt] * (1 if dt == "f64" else 5)
    assert float((re.cpu() - torch.from_numpy(h_re)).abs().max()) < lim and float((im.cpu() - torch.from_numpy(h_im)).abs().max()) < lim
    _guards_intact(bufs, off, n, n)
    # a batch of three, an odd number of elements apart (n + 1 and n + 3), on the same kind of view
    for pad in (1, 3):
        dist, batch = n + pad, 3
        total = (batch - 1) * dist + n
        bufs, (bre, bim) = _views(torch, tdt, off, total, total)
        bre.fill_(GUARD); bim.fill_(GUARD)
        for b in range(batch):
            bre[b * dist:b * dist + n].copy_(torch.from_numpy(h_re)); bim[b * dist:b * dist + n].copy_(torch.from_numpy(h_im))
        gpu.fft_dit_batched(bre, bim, n, gpu.Direction.Forward, planner, dist=dist)
        g_re, g_im = bre.cpu().numpy(), bim.cpu().numpy()
        for b in range(batch):
            sl = slice(b * dist, b * dist + n)
            tol.check(f"unaligned_batch off={off} dist=n+{pad} b={b}", dt, k, g_re[sl], g_im[sl], z.real, z.imag)
            if b + 1 < batch:
                assert np.all(g_re[b * dist + n:(b + 1) * dist] == GUARD) and np.all(g_im[b * dist + n:(b + 1) * dist] == GUARD)
        _guards_intact(bufs, off, total, total)


@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("k", SIZES)
@pytest.mark.parametrize("off", OFFS)
def test_bit_reversal_on_element_aligned_views(gpu, oracle, dt, k, off):
    """phast_bit_rev_{f64,f32}_dev: bit-exact, also where the 16-byte path of bitrev.hip cannot be taken"""
    import torch

    n = 1 << k
    ndt, tdt = (np.float64, torch.float64) if dt == "f64" else (np.float32, torch.float32)
    data = np.arange(n).astype(ndt)
    bufs, (d,) = _views(torch, tdt, off, n)
    d.copy_(torch.from_numpy(data))
    (gpu.bit_rev_bravo_f64 if dt == "f64" else gpu.bit_rev_bravo_f32)(d, k)
    want = data.copy()
    (oracle.bit_rev_bravo_f64 if dt == "f64" else oracle.bit_rev_bravo_f32)(want, k)
    assert np.array_equal(d.cpu().numpy(), want)
    _guards_intact(bufs, off, n)


@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("k", SIZES)
@pytest.mark.parametrize("off", OFFS)
def test_real_transforms_on_element_aligned_views(gpu, oracle, dt, k, off):
    """phast_r2c_fft_*_dev / phast_c2r_fft_*_dev: the first pass of R2C reads the real signal as (even, odd) PAIRS and the last
    pass of C2R stores pairs -- through a pointer aligned to one element only; outputs of N/2 + 1 elements on such views too"""
    import torch

    n = 1 << k
    h1 = n // 2 + 1
    ndt, tdt = (np.float64, torch.float64) if dt == "f64" else (np.float32, torch.float32)
    planner = (gpu.PlannerR2c64 if dt == "f64" else gpu.PlannerR2c32)(n)
    r2c = gpu.r2c_fft_f64_with_planner if dt == "f64" else gpu.r2c_fft_f32_with_planner
    c2r = gpu.c2r_fft_f64_with_planner if dt == "f64" else gpu.c2r_fft_f32_with_planner
    h_x, _ = oracle.fill(n, ndt, seed=0xA12, transform_id=k)
    bufs, (x, ore, oim, back) = _views(torch, tdt, off, n, h1, h1, n)
    x.copy_(torch.from_numpy(h_x))
    r2c(x, ore, oim, planner)
    assert np.array_equal(x.cpu().numpy(), h_x)                  # `&[T]`: the input is not modified (r2c.rs:535)
    ref = np.fft.rfft(h_x.astype(np.float64))
    tol.check(f"unaligned_r2c off={off}", dt, k, ore.cpu().numpy(), oim.cpu().numpy(), ref.real, ref.imag)
    o_re, o_im = np.zeros(h1, ndt), np.zeros(h1, ndt)
    (oracle.r2c_fft_f64 if dt == "f64" else oracle.r2c_fft_f32)(h_x.copy(), o_re, o_im)
    tol.check(f"unaligned_r2c_vs_oracle off={off}", dt, k, ore.cpu().numpy(), oim.cpu().numpy(), o_re.astype(np.float64),
              o_im.astype(np.float64), against="oracle_real")
    c2r(ore, oim, back, planner)
    want = np.zeros(n, ndt)
    (oracle.c2r_fft_f64 if dt == "f64" else oracle.c2r_fft_f32)(ore.cpu().numpy().copy(), oim.cpu().numpy().copy(), want)
    tol.check_real(f"unaligned_c2r_vs_oracle off={off}", dt, k, back.cpu().numpy(), want, against="oracle_real")
    assert float((back.cpu() - torch.from_numpy(h_x)).abs().max()) < tol.ROUNDTRIP_ABS[dt] * (1 if dt == "f64" else 5)
    _guards_intact(bufs, off, n, h1, h1, n)


@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("k", SIZES)
@pytest.mark.parametrize("off", OFFS)
def test_interleaved_on_element_aligned_views(gpu, oracle, dt, k, off):
    """phast_fft_{64,32}_interleaved_dev: `&mut [Complex<T>]` aligned to ONE complex element's scalar (lib.rs:41-140): the view
    starts `off` complex elements into a buffer -- and, through a real view of the same storage, ONE scalar into it"""
    import torch

    n = 1 << k
    ndt, cdt, tdt = (np.float64, np.complex128, torch.complex128) if dt == "f64" else (np.float32, np.complex64, torch.complex64)
    planner = (gpu.PlannerDit64 if dt == "f64" else gpu.PlannerDit32)(n)
    fft = gpu.fft_64_interleaved_with_planner if dt == "f64" else gpu.fft_32_interleaved_with_planner
    h_re, h_im = oracle.fill(n, ndt, seed=0xA13, transform_id=k)
    z0 = (h_re + 1j * h_im).astype(cdt)
    want = np.fft.fft(z0.astype(np.complex128))
    # (a) a view `off` complex elements in
    buf = torch.full((off + n + 3,), GUARD, dtype=tdt, device="cuda")
    v = buf[off:off + n]
    v.copy_(torch.from_numpy(z0))
    fft(v, gpu.Direction.Forward, planner)
    tol.check_c(f"unaligned_interleaved off={off}", dt, k, v.cpu().numpy().astype(np.complex128), want)
    h = buf.cpu().numpy()
    assert np.all(h[:off] == GUARD) and np.all(h[off + n:] == GUARD)
    # (b) a pointer ONE SCALAR into the storage: pairs that straddle the natural 2-scalar alignment.  torch cannot express
    # such a complex view (view_as_complex wants an even storage offset), a C or Rust caller can: straight through the C ABI
    import ctypes as C

    from phastft_amd import _lib

    rdt = torch.float64 if dt == "f64" else torch.float32
    flat = torch.full((2 * n + 2 * off + 9,), GUARD, dtype=rdt, device="cuda")
    start = 2 * off - 1                                            # odd: the pair (re, im) starts on an odd scalar index
    view = flat[start:start + 2 * n]
    assert view.data_ptr() % (16 if dt == "f64" else 8) != 0
    view.copy_(torch.from_numpy(np.ascontiguousarray(z0).view(ndt)))
    sfx = "64" if dt == "f64" else "32"
    rc = getattr(_lib.lib(), f"phast_fft_{sfx}_interleaved_dev")(C.c_void_p(view.data_ptr()), C.c_size_t(n), C.c_size_t(1), C.c_size_t(n),
                                                                  C.c_int(int(gpu.Direction.Forward)), planner._h,
                                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    torch.cuda.synchronize()
    got = view.cpu().numpy().astype(np.float64)
    tol.check_c(f"unaligned_interleaved_scalar off={off}", dt, k, got[0::2] + 1j * got[1::2], want)
    h = flat.cpu().numpy()
    assert np.all(h[:start] == GUARD) and np.all(h[start + 2 * n:] == GUARD)
