"""Complex<T> <-> planes on the device (complex_nums.rs:11-56: `deinterleave`, `deinterleave_complex64 / 32`, `combine_re_im`;
public upstream with feature `bench-internals`, like the bit reversal) -- the data-format step either side of the transform.

Pure data movement: BIT-EXACT against the oracle's restatement, for the reference's own list of lengths
(complex_nums.rs:75: 0, 1, 2, 3, 15, 16, 17, 127 ... 100500 -- odd lengths drop their last element, `chunks_exact(2)`), at
sizes far beyond the caches (2^26 + 1 scalars), on element-aligned views (`&v[1..]`: the 16-byte path cannot be taken), through
the device AND the host-slice forms of the C ABI; the elements around the outputs must come back untouched."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LENGTHS = [0, 1, 2, 3, 15, 16, 17, 127, 128, 129, 130, 135, 100500, (1 << 20), (1 << 26) + 1]
GUARD = 7.0


def _input(n, ndt):
    # distinct values that are exact in f32 too: position modulo 2^24 (what matters is WHERE an element ends up), sign flips
    x = (np.arange(n, dtype=np.int64) % (1 << 24)).astype(ndt)
    x[1::3] *= -1
    return x


@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("n", LENGTHS)
def test_deinterleave_and_combine_device(gpu, oracle, dt, n):
    import torch

    ndt = np.float64 if dt == "f64" else np.float32
    h = _input(n, ndt)
    want_a, want_b = oracle.deinterleave(h)
    d = torch.from_numpy(h).cuda()
    a, b = gpu.deinterleave(d)
    assert a.numel() == n // 2 and b.numel() == n // 2
    assert np.array_equal(a.cpu().numpy(), want_a) and np.array_equal(b.cpu().numpy(), want_b)
    assert np.array_equal(d.cpu().numpy(), h)                       # `&[T]`: the input is not modified
    z = gpu.combine_re_im(a, b)                                     # a complex tensor of n / 2 elements
    assert z.is_complex() and z.numel() == n // 2
    got = torch.view_as_real(z).reshape(-1).cpu().numpy() if n >= 2 else np.empty(0, ndt)
    assert np.array_equal(got, oracle.combine_re_im(want_a, want_b))
    assert np.array_equal(got, h[:2 * (n // 2)])                    # complex_nums.rs:83-118: separate, combine: the input again
    if n >= 2:                                                      # deinterleave_complex64 / 32 on the complex view of the same data
        fn = gpu.deinterleave_complex64 if dt == "f64" else gpu.deinterleave_complex32
        re, im = fn(z)
        assert np.array_equal(re.cpu().numpy(), want_a) and np.array_equal(im.cpu().numpy(), want_b)


@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("n", [0, 1, 2, 3, 17, 130, 100500, 1 << 20])
def test_host_slices(gpu, oracle, dt, n):
    ndt = np.float64 if dt == "f64" else np.float32
    h = _input(n, ndt)
    want_a, want_b = oracle.deinterleave(h)
    a, b = gpu.deinterleave(h)
    assert isinstance(a, np.ndarray) and np.array_equal(a, want_a) and np.array_equal(b, want_b)
    z = gpu.combine_re_im(a, b)
    assert z.dtype == (np.complex128 if dt == "f64" else np.complex64)
    assert np.array_equal(z.view(ndt), h[:2 * (n // 2)])
    cz = h[:2 * (n // 2)].view(np.complex128 if dt == "f64" else np.complex64)
    re, im = (gpu.deinterleave_complex64 if dt == "f64" else gpu.deinterleave_complex32)(cz)
    assert np.array_equal(re, want_a) and np.array_equal(im, want_b)


@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("off", [1, 3])
@pytest.mark.parametrize("n", [16, 131, 100500, (1 << 20) + 6])
def test_element_aligned_views_and_guards(gpu, oracle, dt, off, n):
    """pointers aligned to ONE element (`&v[1..]`): the 16-byte path of complex_nums.hip cannot be taken; nothing outside the
    output slices is written -- straight through the C ABI, every combination of aligned / unaligned streams"""
    import torch

    from phastft_amd import _lib

    ndt, tdt = (np.float64, torch.float64) if dt == "f64" else (np.float32, torch.float32)
    h = _input(n, ndt)
    want_a, want_b = oracle.deinterleave(h)
    pairs = n // 2
    lib, st = _lib.lib(), C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for off_in, off_a, off_b in [(off, 0, 0), (0, off, 0), (0, 0, off), (off, off, off)]:
        src = torch.full((n + off_in + 5,), GUARD, dtype=tdt, device="cuda")
        src[off_in:off_in + n].copy_(torch.from_numpy(h))
        oa = torch.full((pairs + off_a + 5,), GUARD, dtype=tdt, device="cuda")
        ob = torch.full((pairs + off_b + 5,), GUARD, dtype=tdt, device="cuda")
        rc = getattr(lib, f"phast_deinterleave_{dt}_dev")(C.c_void_p(src[off_in:].data_ptr()), C.c_size_t(n), C.c_void_p(oa[off_a:].data_ptr()),
                                                          C.c_void_p(ob[off_b:].data_ptr()), st)
        assert rc == 0
        ha, hb = oa.cpu().numpy(), ob.cpu().numpy()
        assert np.array_equal(ha[off_a:off_a + pairs], want_a) and np.array_equal(hb[off_b:off_b + pairs], want_b)
        assert np.all(ha[:off_a] == GUARD) and np.all(ha[off_a + pairs:] == GUARD)
        assert np.all(hb[:off_b] == GUARD) and np.all(hb[off_b + pairs:] == GUARD)
        # ... and back: combine into a view of the same kind
        out = torch.full((2 * pairs + off_in + 5,), GUARD, dtype=tdt, device="cuda")
        rc = getattr(lib, f"phast_combine_re_im_{dt}_dev")(C.c_void_p(oa[off_a:].data_ptr()), C.c_void_p(ob[off_b:].data_ptr()), C.c_size_t(pairs),
                                                           C.c_void_p(out[off_in:].data_ptr()), st)
        assert rc == 0
        ho = out.cpu().numpy()
        assert np.array_equal(ho[off_in:off_in + 2 * pairs], h[:2 * pairs])
        assert np.all(ho[:off_in] == GUARD) and np.all(ho[off_in + 2 * pairs:] == GUARD)


def test_length_checks(gpu):
    """complex_nums.rs:48 `assert_eq!(reals.len(), imags.len())`; the host-slice forms check the outputs a C caller brings"""
    from phastft_amd import _lib

    with pytest.raises(gpu.PhastPanic):
        gpu.combine_re_im(np.zeros(4), np.zeros(5))
    lib = _lib.lib()
    x, a, b = np.zeros(10), np.zeros(5), np.zeros(4)
    p = lambda v: v.ctypes.data_as(C.c_void_p)
    assert lib.phast_deinterleave_f64(p(x), C.c_size_t(10), p(a), C.c_size_t(5), p(b), C.c_size_t(4)) == 2   # PHAST_ERR_LEN_MISMATCH
    assert lib.phast_combine_re_im_f64(p(a), C.c_size_t(5), p(b), C.c_size_t(4), p(x), C.c_size_t(10)) == 2
    assert lib.phast_combine_re_im_f64(p(a), C.c_size_t(5), p(a), C.c_size_t(5), p(x), C.c_size_t(9)) == 2
