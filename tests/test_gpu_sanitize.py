"""Device-side out-of-bounds check of the pass kernels (SURVEY.md section 5 "race detection / sanitizers"; VERDICT r02
item 9).  The GPU boxes run gfx950 with xnack off, so ASan's device instrumentation is not available; instead every
buffer a kernel may write is wrapped in GUARD BANDS:

  * the caller's planes sit in the middle of larger tensors whose margins hold a canary value -- bit-exact afterwards;
  * the planner's scratch is allocated with 1 MiB bands of 0xA5 on either side (phast_debug_set_guard_bytes) and
    `planner.check_guards()` counts overwritten bytes.

Workloads: every plan family (one-pass small transforms, latency / mid / single (wave + quad) / throughput plans, the
tw3_global passes of 2^28), ragged batches (dist > n), strided batches, the interleaved API, R2C / C2R, bit reversal.
The host side of the same pass is tools/sanitize_host.sh (libphastft_hip.so and the C++ host test under ASan;
log in profiles/).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CANARY = 1234.5
PAD = 1 << 16  # elements of margin on either side of every user buffer


def guarded(torch, count, dtype):
    """(whole, view): `count` elements in the middle of a canary-filled tensor"""
    whole = torch.full((count + 2 * PAD,), CANARY, dtype=dtype, device="cuda")
    return whole, whole[PAD:PAD + count]


def margins_intact(torch, whole, count):
    ok_lo = bool((whole[:PAD] == CANARY).all())
    ok_hi = bool((whole[PAD + count:] == CANARY).all())
    return ok_lo and ok_hi


@pytest.fixture(scope="module")
def guards(gpu):
    gpu.debug_set_guard_bytes(1 << 20)
    yield gpu
    gpu.debug_set_guard_bytes(0)


@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("k,batch,dist_extra", [(5, 1000, 0), (9, 77, 3), (13, 33, 64), (14, 37, 0), (16, 5, 128), (18, 3, 0),
                                                (20, 1, 0), (20, 2, 0), (20, 8, 64), (20, 64, 0), (21, 1, 0), (22, 1, 0),
                                                (23, 1, 0), (24, 1, 0), (24, 4, 0)])
def test_c2c_writes_stay_inside_the_callers_planes_and_the_scratch(guards, k, batch, dist_extra, dt):
    import torch

    gpu = guards
    n = 1 << k
    dist = n + dist_extra
    tdt = torch.float64 if dt == "f64" else torch.float32
    planner = (gpu.PlannerDit64 if dt == "f64" else gpu.PlannerDit32)(n)
    count = dist * (batch - 1) + n
    w_re, re = guarded(torch, count, tdt)
    w_im, im = guarded(torch, count, tdt)
    re.uniform_(-1, 1)
    im.uniform_(-1, 1)
    gaps_before = None
    if dist_extra:  # the gaps between ragged transforms must not be touched either
        idx = torch.arange(count, device="cuda") % dist >= n
        gaps_before = (re[idx].clone(), im[idx].clone())
    for direction in (gpu.Direction.Forward, gpu.Direction.Reverse):
        gpu.fft_dit_batched(re, im, n, direction, planner, dist=dist)
    torch.cuda.synchronize()
    assert margins_intact(torch, w_re, count) and margins_intact(torch, w_im, count), planner.describe()
    assert planner.check_guards() == 0, planner.describe()
    if gaps_before is not None:
        assert torch.equal(re[idx], gaps_before[0]) and torch.equal(im[idx], gaps_before[1])


def test_tw3_global_passes_2p28(guards):
    import torch

    gpu = guards
    n = 1 << 28
    planner = gpu.PlannerDit64(n)
    w_re, re = guarded(torch, n, torch.float64)
    w_im, im = guarded(torch, n, torch.float64)
    gpu.fill_uniform(re, im, n)
    gpu.fft_64_dit_with_planner(re, im, gpu.Direction.Forward, planner)
    torch.cuda.synchronize()
    assert margins_intact(torch, w_re, n) and margins_intact(torch, w_im, n)
    assert planner.check_guards() == 0, planner.describe()


@pytest.mark.parametrize("k,s,sb,dt", [(8, 5, 5, "f64"), (12, 6, 5, "f64"), (16, 4, 4, "f64"), (14, 6, 6, "f32"), (20, 5, 5, "f32")])
def test_strided_batches(guards, k, s, sb, dt):
    import torch

    gpu = guards
    n, stride, batch = 1 << k, 1 << s, 1 << sb
    tdt = torch.float64 if dt == "f64" else torch.float32
    planner = (gpu.PlannerDit64 if dt == "f64" else gpu.PlannerDit32)(n)
    w_re, re = guarded(torch, n * stride, tdt)
    w_im, im = guarded(torch, n * stride, tdt)
    re.uniform_(-1, 1)
    im.uniform_(-1, 1)
    untouched = re.view(n, stride)[:, batch:].clone() if batch < stride else None
    gpu.fft_dit_strided(re, im, n, gpu.Direction.Forward, planner, batch=batch, stride=stride)
    torch.cuda.synchronize()
    assert margins_intact(torch, w_re, n * stride) and margins_intact(torch, w_im, n * stride)
    assert planner.check_guards() == 0
    if untouched is not None:
        assert torch.equal(re.view(n, stride)[:, batch:], untouched)


@pytest.mark.parametrize("k", [6, 12, 15, 20, 24])
def test_real_transforms_interleaved_and_bit_reversal(guards, k):
    import torch

    gpu = guards
    n = 1 << k
    for dt, R2C, r2c, c2r in ((torch.float64, gpu.PlannerR2c64, gpu.r2c_fft_f64_with_planner, gpu.c2r_fft_f64_with_planner),
                              (torch.float32, gpu.PlannerR2c32, gpu.r2c_fft_f32_with_planner, gpu.c2r_fft_f32_with_planner)):
        pl = R2C(n)
        w_x, x = guarded(torch, n, dt)
        w_a, a = guarded(torch, n // 2 + 1, dt)
        w_b, b = guarded(torch, n // 2 + 1, dt)
        x.uniform_(-1, 1)
        x0 = x.clone()
        r2c(x, a, b, pl)
        torch.cuda.synchronize()
        assert torch.equal(x, x0), "the input of r2c is read-only (r2c.rs:535)"
        c2r(a, b, x, pl)
        torch.cuda.synchronize()
        for w, c in ((w_x, n), (w_a, n // 2 + 1), (w_b, n // 2 + 1)):
            assert margins_intact(torch, w, c), (k, dt)
        assert float((x - x0).abs().max()) < (1e-9 if dt == torch.float64 else 1e-3)
        # interleaved complex signal of n points and the stand-alone bit reversal
        w_z, z = guarded(torch, 2 * n, dt)
        z.uniform_(-1, 1)
        zc = torch.view_as_complex(z.view(n, 2))
        (gpu.fft_64_interleaved if dt == torch.float64 else gpu.fft_32_interleaved)(zc, gpu.Direction.Forward)
        w_v, v = guarded(torch, n, dt)
        v.uniform_(-1, 1)
        (gpu.bit_rev_bravo_f64 if dt == torch.float64 else gpu.bit_rev_bravo_f32)(v, k)
        torch.cuda.synchronize()
        assert margins_intact(torch, w_z, 2 * n) and margins_intact(torch, w_v, n), (k, dt)
