"""GPU parity tests proper: the HIP path, called through the C ABI (phastft_amd -> libphastft_hip.so),
against the CPU oracle on the same seeded inputs.  Ports the reference's own tests
(lib.rs:238-461, bravo.rs:373-407, r2c.rs:914-1540) and adds the BASELINE.json sizes.

Tolerances: bit reversal exact; everything else through tests/tolerances.py (round 6: every comparison) -- gates tied to
the measured error of the HIP path: f64 rel-L2 <= 8e-16 log2 N and worst bin <= 64 eps log2 N rms; f32 against a float64
reference 1.5e-7 log2 N / 2e-6 log2 N rms, against the f32 ORACLE the documented loose bound that absorbs the reference's
3.5-ulp f32 planner twiddles; f64 R2C / C2R against the oracle the bound that absorbs its rotation-recurrence drift AND the
f64 formula against an independent real FFT; round trips: the reference's own absolute bounds (lib.rs:398,421).
"""
import numpy as np
import pytest

from tests import tolerances as tol

pytestmark = pytest.mark.gpu


def rel_l2(got_re, got_im, ref_re, ref_im):
    num = np.sqrt(np.sum((got_re.astype(np.float64) - ref_re) ** 2 + (got_im.astype(np.float64) - ref_im) ** 2))
    den = np.sqrt(np.sum(ref_re.astype(np.float64) ** 2 + ref_im.astype(np.float64) ** 2))
    return num / den if den else num


def max_bin_err(got_re, got_im, ref_re, ref_im):
    """largest single-bin error relative to the rms bin magnitude (round 4): a few wrong bins move
    the rel-L2 by ~sqrt(bins/N) only -- this bound catches them outright"""
    e = np.maximum(np.abs(got_re.astype(np.float64) - ref_re), np.abs(got_im.astype(np.float64) - ref_im))
    rms = np.sqrt(np.mean(np.asarray(ref_re, np.float64) ** 2 + np.asarray(ref_im, np.float64) ** 2))
    return float(e.max()) / rms if rms else float(e.max())



def dev(x):
    import torch

    return torch.from_numpy(x).cuda()


# ---------------------------------------------------------------- bit reversal (bravo.rs:373-407)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_bit_reversal_exact_all_regimes(gpu, oracle, dtype):
    fn = gpu.bit_rev_bravo_f64 if dtype == np.float64 else gpu.bit_rev_bravo_f32
    ofn = oracle.bit_rev_bravo_f64 if dtype == np.float64 else oracle.bit_rev_bravo_f32
    for n in range(0, 24):
        big_n = 1 << n
        data = np.arange(big_n).astype(dtype)  # data[i] = i as the reference's test
        d = dev(data.copy())
        fn(d, n)
        expect = data.copy()
        if n >= 1:
            ofn(expect, n)
        got = d.cpu().numpy()
        assert np.array_equal(got.view(np.uint64 if dtype == np.float64 else np.uint32),
                              expect.view(np.uint64 if dtype == np.float64 else np.uint32)), n
    # host-slice entry point, and involution on random bits incl. NaN payloads
    rng = np.random.default_rng(7)
    for n in (3, 10, 13):
        raw = rng.integers(0, 2**63, size=1 << n, dtype=np.uint64)
        data = raw.view(np.float64).copy() if dtype == np.float64 else raw.view(np.uint32)[: 1 << n].view(np.float32).copy()
        orig = data.copy()
        fn(data, n)
        fn(data, n)
        assert np.array_equal(data.view(np.uint8), orig.view(np.uint8))


def test_bit_reversal_large_involution_and_checksum(gpu):
    import torch

    n = 26
    x = torch.arange(1 << n, dtype=torch.float64, device="cuda")
    gpu.bit_rev_bravo_f64(x, n)
    # spot-check the permutation: x[i] == rev(i) for sampled i, and the multiset is preserved
    idx = torch.tensor([0, 1, 2, 3, 5, (1 << n) - 1, (1 << 25) + 12345, 987654], device="cuda")
    rev = torch.zeros_like(idx)
    for b in range(n):
        rev |= ((idx >> b) & 1) << (n - 1 - b)
    assert torch.equal(x[idx].to(torch.int64), rev)
    assert float(x.sum()) == float((1 << n) * ((1 << n) - 1) // 2)
    gpu.bit_rev_bravo_f64(x, n)
    assert torch.equal(x, torch.arange(1 << n, dtype=torch.float64, device="cuda"))


@pytest.mark.parametrize("n", [27, 28])
def test_bit_reversal_128x128_register_tiles_full_permutation(gpu, n):
    """2^27 points and up run the third-generation kernel (128 x 128 tiles through registers and one LDS buffer, the next
    pair prefetched): the WHOLE permutation against integer index arithmetic, then the involution."""
    import torch

    x = torch.arange(1 << n, dtype=torch.float64, device="cuda")
    gpu.bit_rev_bravo_f64(x, n)
    i = torch.arange(1 << n, dtype=torch.int64, device="cuda")
    rev = torch.zeros_like(i)
    for b in range(n):
        rev |= ((i >> b) & 1) << (n - 1 - b)
    del i
    assert torch.equal(x.to(torch.int64), rev)
    del rev
    gpu.bit_rev_bravo_f64(x, n)
    assert torch.equal(x, torch.arange(1 << n, dtype=torch.float64, device="cuda"))


# ---------------------------------------------------------------- C2C vs the oracle
@pytest.mark.parametrize("k", list(range(0, 23)))
def test_fft_64_vs_oracle(gpu, oracle, k):
    n = 1 << k
    re, im = oracle.fill(n, np.float64, seed=0xCAFE, transform_id=k)
    d_re, d_im = dev(re.copy()), dev(im.copy())
    gpu.fft_64_dit(d_re, d_im, gpu.Direction.Forward)
    oracle.fft_64_dit(re, im, oracle.FORWARD)
    # round 5: the gates of tests/tolerances.py (rel-L2 <= 8e-16 log2 N, worst bin <= 64 eps log2 N rms)
    tol.check("c2c_vs_oracle", "f64", k, d_re.cpu().numpy(), d_im.cpu().numpy(), re, im)


@pytest.mark.parametrize("k", list(range(0, 23)))
def test_fft_32_vs_oracle(gpu, oracle, k):
    n = 1 << k
    re, im = oracle.fill(n, np.float32, seed=0xCAFE, transform_id=k)
    d_re, d_im = dev(re.copy()), dev(im.copy())
    gpu.fft_32_dit(d_re, d_im, gpu.Direction.Forward)
    ref = np.fft.fft(re.astype(np.float64) + 1j * im.astype(np.float64))  # (before the oracle transforms re, im in place)
    oracle.fft_32_dit(re, im, oracle.FORWARD)
    g_re, g_im = d_re.cpu().numpy(), d_im.cpu().numpy()
    tol.check("c2c_vs_oracle", "f32", k, g_re, g_im, re, im, against="oracle")   # the loose bound: absorbs the reference's f32 twiddles
    tol.check("c2c_vs_f64", "f32", k, g_re, g_im, ref.real, ref.imag)            # nothing to absorb: 1.5e-7 log2 N


def test_fft_correctness_ramp_like_reference(gpu):
    """lib.rs:298-338: re = im = 1..N vs an independent FFT, abs 0.01 (f64 N=2^4..2^16, f32 2^4..2^8)."""
    for k in range(4, 17):
        n = 1 << k
        ramp = np.arange(1, n + 1, dtype=np.float64)
        re, im = ramp.copy(), ramp.copy()
        gpu.fft_64_dit(re, im, gpu.Direction.Forward)  # host slices
        ref = np.fft.fft(ramp + 1j * ramp)
        assert np.max(np.abs(re - ref.real)) < 0.01 and np.max(np.abs(im - ref.imag)) < 0.01, k
    for k in range(4, 9):
        n = 1 << k
        ramp = np.arange(1, n + 1, dtype=np.float32)
        re, im = ramp.copy(), ramp.copy()
        gpu.fft_32_dit(re, im, gpu.Direction.Forward)
        ref = np.fft.fft(ramp.astype(np.float64) * (1 + 1j))
        assert np.max(np.abs(re - ref.real)) < 0.01 and np.max(np.abs(im - ref.imag)) < 0.01, k


def test_roundtrip_like_reference(gpu):
    """lib.rs:381-425: forward then inverse on unit-norm random signals, 1e-10 / 1e-7."""
    rng = np.random.default_rng(3)
    for k in range(4, 12):
        n = 1 << k
        re, im = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
        s = 1.0 / np.sqrt(np.sum(re * re + im * im))
        re, im = re * s, im * s
        r, m = re.copy(), im.copy()
        gpu.fft_64_dit(r, m, gpu.Direction.Forward)
        gpu.fft_64_dit(r, m, gpu.Direction.Reverse)
        assert np.max(np.abs(r - re)) < 1e-10 and np.max(np.abs(m - im)) < 1e-10
        r32, m32 = re.astype(np.float32), im.astype(np.float32)
        o_r, o_m = r32.copy(), m32.copy()
        gpu.fft_32_dit(r32, m32, gpu.Direction.Forward)
        gpu.fft_32_dit(r32, m32, gpu.Direction.Reverse)
        assert np.max(np.abs(r32 - o_r)) < 1e-7 and np.max(np.abs(m32 - o_m)) < 1e-7


def test_inverse_matches_oracle(gpu, oracle):
    for k in (3, 9, 14, 20):
        n = 1 << k
        re, im = oracle.fill(n, np.float64, transform_id=100 + k)
        d_re, d_im = dev(re.copy()), dev(im.copy())
        gpu.fft_64_dit(d_re, d_im, gpu.Direction.Reverse)
        oracle.fft_64_dit(re, im, oracle.REVERSE)
        tol.check("inverse_vs_oracle", "f64", k, d_re.cpu().numpy(), d_im.cpu().numpy(), re, im)


@pytest.mark.parametrize("plan", [((10, 10), 13, 4), ((7, 7, 6), 12, 4), ((8, 6, 6), 12, 4), ((6, 6, 8), 12, 4), ((10, 10), 14, 5),
                                  ((10, 10), 14, 4), ((10, 10), 13, 5), ((10, 10), 12, 5), ((6, 8, 6), 12, 3),
                                  ((10, 10), 12, 3),
                                  # small tiles (several workgroups per CU) and WAVE tiles (points code | 0x10: every
                                  # 64 x 16 tile is one wave, v_permlane swaps instead of LDS exchanges -- wave_fft.hpp)
                                  ((7, 6, 7), (11, 10, 11), 3), ((6, 7, 7), (10, 11, 11), 4),
                                  ((6, 8, 6), (10, 12, 10), 3 | 0x10), ((6, 8, 6), (10, 12, 10), 4 | 0x10),
                                  ((7, 6, 7), (11, 10, 11), 3 | 0x10), ((6, 7, 7), (10, 11, 11), 4 | 0x10)])
def test_forced_plans_agree_2p20(gpu, oracle, plan):
    n = 1 << 20
    lrs, tl, lp = plan
    planner = gpu.PlannerDit64(n)
    planner.set_plan(lrs, tl, lp)  # every plan of the list exists as kernels: a refusal is a regression of set_plan
    re, im = oracle.fill(n, np.float64, transform_id=5)
    d_re, d_im = dev(re.copy()), dev(im.copy())
    gpu.fft_64_dit_with_planner(d_re, d_im, gpu.Direction.Forward, planner)
    oracle.fft_64_dit(re, im, oracle.FORWARD)
    tol.check("forced_plan_2p20 " + planner.describe_call(), "f64", 20, d_re.cpu().numpy(), d_im.cpu().numpy(), re, im)


def test_config3_2p26_roundtrip_and_sampled_bins(gpu):
    """BASELINE config 3: N=2^26 f64 forward+inverse on 1 GPU; size-independent properties instead of a
    full oracle run: (a) sampled output bins vs a direct O(N) long-double-free DFT in f64 with compensated
    phase, (b) Parseval, (c) round trip <= 1e-10 * scale."""
    import torch

    n = 1 << 26
    re = torch.empty(n, dtype=torch.float64, device="cuda")
    im = torch.empty(n, dtype=torch.float64, device="cuda")
    gpu.fill_uniform(re, im, n, seed=0xCAFE, first_id=0)
    re0, im0 = re.clone(), im.clone()
    planner = gpu.PlannerDit64(n)
    gpu.fft_64_dit_with_planner(re, im, gpu.Direction.Forward, planner)
    # (b) Parseval: sum |X|^2 = N sum |x|^2
    e_in = float((re0 * re0 + im0 * im0).sum())
    e_out = float((re * re + im * im).sum())
    assert abs(e_out / (n * e_in) - 1.0) < 1e-12
    # (a) direct DFT of sampled bins; the phase k*j/N is reduced exactly in integers before the sincos
    j = torch.arange(n, dtype=torch.int64, device="cuda")
    for k in (0, 1, 12345, n // 2, n - 1, 3 * (n // 8) + 7):
        ph = (j * k) % n
        ang = ph.to(torch.float64) * (-2.0 * np.pi / n)
        c, s = torch.cos(ang), torch.sin(ang)
        xr = float((re0 * c - im0 * s).sum())
        xi = float((re0 * s + im0 * c).sum())
        scale = np.sqrt(n * e_in)  # ||X||_2
        gate = tol.f64_bin(26) * scale / np.sqrt(n)   # per bin, relative to the rms bin = ||X|| / sqrt(N)
        assert abs(float(re[k]) - xr) < gate and abs(float(im[k]) - xi) < gate, k
    del j, ph, ang, c, s
    # (c) round trip
    gpu.fft_64_dit_with_planner(re, im, gpu.Direction.Reverse, planner)
    assert float((re - re0).abs().max()) < 1e-10 and float((im - im0).abs().max()) < 1e-10


def test_batched_matches_single(gpu, oracle):
    import torch

    n, batch = 1 << 14, 37
    re = torch.empty(n * batch, dtype=torch.float64, device="cuda")
    im = torch.empty_like(re)
    gpu.fill_uniform(re, im, n, seed=0xCAFE, first_id=1000)
    planner = gpu.PlannerDit64(n)
    gpu.fft_dit_batched(re, im, n, gpu.Direction.Forward, planner)
    for b in (0, 1, 17, 36):
        r, m = oracle.fill(n, np.float64, seed=0xCAFE, transform_id=1000 + b)
        oracle.fft_64_dit(r, m, oracle.FORWARD)
        tol.check("batched_vs_oracle", "f64", n.bit_length() - 1, re[b * n:(b + 1) * n].cpu().numpy(), im[b * n:(b + 1) * n].cpu().numpy(), r, m)


@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("k", list(range(12, 24)))
def test_throughput_plans_vs_oracle(gpu, oracle, k, dt):
    """A single small transform runs the latency plan; this drives the THROUGHPUT plan of every size (2^25 points
    in flight is past every crossover of plan.hpp: throughput_work): first / second / last transform against the
    oracle, every transform through Parseval."""
    import torch

    n = 1 << k
    batch = (1 << 25) // n
    tdt, ndt = (torch.float64, np.float64) if dt == "f64" else (torch.float32, np.float32)
    planner = (gpu.PlannerDit64 if dt == "f64" else gpu.PlannerDit32)(n)
    re = torch.empty(n * batch, dtype=tdt, device="cuda")
    im = torch.empty_like(re)
    gpu.fill_uniform(re, im, n, seed=0xBEEF, first_id=7)
    e_in = (re.double() ** 2 + im.double() ** 2).view(batch, n).sum(dim=1)
    gpu.fft_dit_batched(re, im, n, gpu.Direction.Forward, planner)
    e_out = (re.double() ** 2 + im.double() ** 2).view(batch, n).sum(dim=1)
    assert float((e_out / (n * e_in) - 1.0).abs().max()) < tol.parseval_gate(dt, k), planner.describe()
    ofn = oracle.fft_64_dit if dt == "f64" else oracle.fft_32_dit
    for b in (0, 1, batch - 1):
        r, m = oracle.fill(n, ndt, seed=0xBEEF, transform_id=7 + b)
        x64 = r.astype(np.float64) + 1j * m.astype(np.float64)
        ofn(r, m, oracle.FORWARD)
        sl = slice(b * n, (b + 1) * n)
        g_re, g_im = re[sl].cpu().numpy(), im[sl].cpu().numpy()
        tol.check("throughput_plan_vs_oracle " + planner.describe_call(batch), dt, k, g_re, g_im, r, m, against="oracle")
        if dt == "f32":   # ... and against float64 pocketfft, where nothing of the reference's needs absorbing
            tol.check_c("throughput_plan_vs_pocketfft", dt, k, g_re.astype(np.float64) + 1j * g_im.astype(np.float64), np.fft.fft(x64))


# ---------------------------------------------------------------- planner misuse (lib.rs:238-296)
def test_panics_like_reference(gpu):
    with pytest.raises(gpu.PhastPanic):
        gpu.PlannerDit64(5)
    with pytest.raises(gpu.PhastPanic):
        gpu.PlannerDit32(5)
    planner = gpu.PlannerDit64(16)
    re, im = np.zeros(1 << 16), np.zeros(1 << 16)
    with pytest.raises(gpu.PhastPanic):
        gpu.fft_64_dit_with_planner_and_opts(re, im, gpu.Direction.Forward, planner, gpu.Options.guess_options(re.size))
    with pytest.raises(gpu.PhastPanic):
        gpu.fft_64_dit(np.zeros(8), np.zeros(16), gpu.Direction.Forward)
    for n in range(5, 15):  # tune_mode_does_not_panic, lib.rs:428-437
        gpu.PlannerDit64.with_mode(1 << n, gpu.PlannerMode.Tune)
        gpu.PlannerDit32.with_mode(1 << n, gpu.PlannerMode.Tune)


# ---------------------------------------------------------------- R2C / C2R (r2c.rs tests)
@pytest.mark.parametrize("k", list(range(2, 21)))
def test_r2c_f64_vs_oracle_and_c2c(gpu, oracle, k):
    n = 1 << k
    x, _ = oracle.fill(n, np.float64, transform_id=k)
    ore, oim = np.full(n // 2 + 1, 7.0), np.full(n // 2 + 1, 7.0)
    gpu.r2c_fft_f64(x, ore, oim)
    ref_re, ref_im = np.zeros(n // 2 + 1), np.zeros(n // 2 + 1)
    oracle.r2c_fft_f64(x, ref_re, ref_im)
    # the oracle reproduces the reference's rotation-recurrence twiddles (planner.rs:128-138), which drift by
    # ~1e-12 (N=2^16) .. 3e-10 (N=2^24); the GPU uses correctly rounded ones, so the bound vs the oracle is the
    # drift, and the tight bound is against an independent real FFT
    tol.check("r2c_f64_vs_oracle", "f64", k, ore, oim, ref_re, ref_im, against="oracle_real")   # its recurrence drift, bin by bin
    ind = np.fft.rfft(x)
    tol.check("r2c_f64_vs_rfft", "f64", k, ore, oim, ind.real, ind.imag)   # no single bin off (exact twiddles on both sides)
    assert oim[0] == 0 and oim[-1] == 0                                # r2c.rs:161-166: exact zeros
    out = np.zeros(n)
    gpu.c2r_fft_f64(ore, oim, out)
    assert np.max(np.abs(out - x)) < 1e-6  # r2c.rs:958-976


@pytest.mark.parametrize("k", list(range(2, 21)))
def test_r2c_f32_vs_oracle(gpu, oracle, k):
    n = 1 << k
    x, _ = oracle.fill(n, np.float32, transform_id=k)
    ore, oim = np.zeros(n // 2 + 1, np.float32), np.zeros(n // 2 + 1, np.float32)
    gpu.r2c_fft_f32(x, ore, oim)
    ref_re, ref_im = np.zeros(n // 2 + 1, np.float32), np.zeros(n // 2 + 1, np.float32)
    oracle.r2c_fft_f32(x, ref_re, ref_im)
    tol.check("r2c_f32_vs_oracle", "f32", k, ore, oim, ref_re.astype(np.float64), ref_im.astype(np.float64), against="oracle")
    ind = np.fft.rfft(x.astype(np.float64))
    tol.check("r2c_f32_vs_rfft", "f32", k, ore, oim, ind.real, ind.imag)
    assert oim[0] == 0 and oim[-1] == 0
    out = np.zeros(n, np.float32)
    gpu.c2r_fft_f32(ore, oim, out)
    assert np.max(np.abs(out - x)) < 5 * tol.ROUNDTRIP_ABS["f32"]


def test_config4_r2c_f32_2p24(gpu, oracle):
    """BASELINE config 4: r2c_fft_f32 at N=2^24 on device tensors vs the oracle."""
    import torch

    n = 1 << 24
    x = torch.empty(n, dtype=torch.float32, device="cuda")
    gpu.fill_uniform(x, None, n, seed=0xCAFE, first_id=4)
    ore = torch.empty(n // 2 + 1, dtype=torch.float32, device="cuda")
    oim = torch.empty_like(ore)
    planner = gpu.PlannerR2c32(n)
    gpu.r2c_fft_f32_with_planner(x, ore, oim, planner)
    hx, _ = oracle.fill(n, np.float32, seed=0xCAFE, transform_id=4)
    assert np.array_equal(hx, x.cpu().numpy())  # the two generators are bit-identical
    ref_re, ref_im = np.zeros(n // 2 + 1, np.float32), np.zeros(n // 2 + 1, np.float32)
    oracle.r2c_fft_f32(hx, ref_re, ref_im)
    g_re, g_im = ore.cpu().numpy(), oim.cpu().numpy()
    tol.check("config4_r2c_f32_2p24_vs_oracle", "f32", 24, g_re, g_im, ref_re.astype(np.float64), ref_im.astype(np.float64), against="oracle")
    ind = np.fft.rfft(hx.astype(np.float64))
    # against float64 pocketfft nothing needs absorbing (round 5): 1.5e-7 log2 N / 2e-6 log2 N rms, every bin of the 2^23 + 1
    tol.check("config4_r2c_f32_2p24", "f32", 24, g_re, g_im, ind.real, ind.imag)
    assert g_im[0] == 0 and g_im[-1] == 0
    back = torch.empty(n, dtype=torch.float32, device="cuda")
    gpu.c2r_fft_f32_with_planner(ore, oim, back, planner)
    assert float((back - x).abs().max()) < 1e-4


def test_r2c_known_answers(gpu):
    """r2c.rs:1235-1386: dc_only, nyquist_only, single_tone, all_zeros (pre-filled outputs), dc_and_nyquist_real."""
    for dtype, fn, tol in ((np.float64, gpu.r2c_fft_f64, 1e-10), (np.float32, gpu.r2c_fft_f32, 1e-4)):
        n, half = 16, 8
        ore, oim = np.zeros(half + 1, dtype), np.zeros(half + 1, dtype)
        fn(np.ones(n, dtype), ore, oim)
        assert abs(ore[0] - n) < tol and np.all(np.abs(ore[1:]) < tol) and np.all(np.abs(oim) < tol)
        x = np.array([1.0 if i % 2 == 0 else -1.0 for i in range(n)], dtype)
        fn(x, ore, oim)
        assert abs(ore[half] - n) < tol and np.all(np.abs(ore[:half]) < tol) and np.all(np.abs(oim) < tol)
        n, half = 32, 16
        ore, oim = np.zeros(half + 1, dtype), np.zeros(half + 1, dtype)
        fn(np.cos(2 * np.pi * np.arange(n) / n).astype(dtype), ore, oim)
        exp = np.zeros(half + 1)
        exp[1] = n / 2
        assert np.all(np.abs(ore - exp) < 10 * tol) and np.all(np.abs(oim) < 10 * tol)
        n, half = 16, 8
        ore, oim = np.ones(half + 1, dtype), np.ones(half + 1, dtype)
        fn(np.zeros(n, dtype), ore, oim)
        assert np.all(ore == 0) and np.all(oim == 0)
        n, half = 64, 32
        ore, oim = np.zeros(half + 1, dtype), np.zeros(half + 1, dtype)
        fn(np.arange(1, n + 1).astype(dtype), ore, oim)
        assert abs(oim[0]) < tol and abs(oim[half]) < tol


def test_r2c_panic_messages(gpu):
    """r2c.rs:1392-1540: exact panic strings."""
    P = gpu
    cases = [
        (lambda: P.PlannerR2c64(6), "n must be a power of 2 >= 4"),
        (lambda: P.PlannerR2c32(2), "n must be a power of 2 >= 4"),
        (lambda: P.r2c_fft_f64(np.zeros(16), np.zeros(8), np.zeros(9)), "output_re must have length N/2 + 1"),
        (lambda: P.r2c_fft_f64(np.zeros(16), np.zeros(9), np.zeros(8)), "output_im must have length N/2 + 1"),
        (lambda: P.r2c_fft_f32_with_planner(np.zeros(8, np.float32), np.zeros(9, np.float32), np.zeros(9, np.float32),
                                            P.PlannerR2c32(16)), "input length must match planner size"),
        (lambda: P.c2r_fft_f64(np.zeros(8), np.zeros(9), np.zeros(16)), "input_re must have length N/2 + 1"),
        (lambda: P.c2r_fft_f64(np.zeros(9), np.zeros(8), np.zeros(16)), "input_im must have length N/2 + 1"),
        (lambda: P.c2r_fft_f64_with_planner(np.zeros(9), np.zeros(9), np.zeros(8), P.PlannerR2c64(16)),
         "output length must match planner size"),
        (lambda: P.c2r_fft_f64_with_planner_and_scratch(np.zeros(9), np.zeros(9), np.zeros(16), P.PlannerR2c64(16),
                                                        np.zeros(7), np.zeros(8)), "scratch_re must have length N/2"),
        (lambda: P.c2r_fft_f64_with_planner_and_scratch(np.zeros(9), np.zeros(9), np.zeros(16), P.PlannerR2c64(16),
                                                        np.zeros(8), np.zeros(7)), "scratch_im must have length N/2"),
    ]
    for fn, msg in cases:
        with pytest.raises(P.PhastPanic) as ei:
            fn()
        assert str(ei.value) == msg


def test_determinism_across_entry_points(gpu):
    """r2c.rs:978-1131: planner vs convenience entry points give bit-identical results."""
    n = 1024
    x = np.arange(1, n + 1, dtype=np.float64)
    a_re, a_im = np.zeros(n // 2 + 1), np.zeros(n // 2 + 1)
    b_re, b_im = np.zeros(n // 2 + 1), np.zeros(n // 2 + 1)
    gpu.r2c_fft_f64(x, a_re, a_im)
    gpu.r2c_fft_f64_with_planner(x, b_re, b_im, gpu.PlannerR2c64(n))
    assert np.array_equal(a_re, b_re) and np.array_equal(a_im, b_im)
    re1, im1 = x.copy(), x.copy()
    re2, im2 = x.copy(), x.copy()
    gpu.fft_64_dit(re1, im1, gpu.Direction.Forward)
    gpu.fft_64_dit_with_planner(re2, im2, gpu.Direction.Forward, gpu.PlannerDit64(n))
    assert np.array_equal(re1, re2) and np.array_equal(im1, im2)
    d1, d2 = dev(x.copy()), dev(x.copy())
    gpu.fft_64_dit(d1, d2, gpu.Direction.Forward)  # device path == host path
    assert np.array_equal(d1.cpu().numpy(), re1) and np.array_equal(d2.cpu().numpy(), im1)


# ---------------------------------------------------------------- interleaved Complex<T> API (lib.rs:41-140, 340-378)
def test_interleaved_matches_planar(gpu):
    """fft_interleaved_correctness (lib.rs:340-378): interleaved == planar at 1e-10, plus inverse and device tensors."""
    import torch

    rng = np.random.default_rng(12)
    for k, cdt, fdt, fwd, planar, tol in ((10, np.complex128, np.float64, gpu.fft_64_interleaved, gpu.fft_64_dit, 1e-10),
                                          (10, np.complex64, np.float32, gpu.fft_32_interleaved, gpu.fft_32_dit, 1e-4),
                                          (16, np.complex128, np.float64, gpu.fft_64_interleaved, gpu.fft_64_dit, 1e-9),
                                          (21, np.complex128, np.float64, gpu.fft_64_interleaved, gpu.fft_64_dit, 1e-8)):
        n = 1 << k
        sig = (np.arange(1, n + 1) + 0j).astype(cdt) if k == 10 else (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(cdt)
        re, im = sig.real.astype(fdt).copy(), sig.imag.astype(fdt).copy()
        orig = sig.copy()
        fwd(sig, gpu.Direction.Forward)
        planar(re, im, gpu.Direction.Forward)
        scale = max(1.0, float(np.max(np.abs(re))))
        assert np.max(np.abs(sig.real - re)) < tol * scale and np.max(np.abs(sig.imag - im)) < tol * scale, k
        d = torch.from_numpy(orig.copy()).cuda()
        fwd(d, gpu.Direction.Forward)
        assert np.array_equal(d.cpu().numpy(), sig)  # device path == host path, bit for bit
        fwd(sig, gpu.Direction.Reverse)
        assert np.max(np.abs(sig - orig)) < (1e-9 if fdt == np.float64 else 1e-3) * max(1.0, float(np.max(np.abs(orig)))), k


@pytest.mark.parametrize("k,dt", [(24, "f64"), (26, "f64"), (27, "f64"), (28, "f64"), (29, "f64"), (24, "f32"),
                                  (25, "f32"), (26, "f32"), (27, "f32"), (29, "f32"), (30, "f32")])
def test_large_sizes_properties(gpu, k, dt):
    """Sizes past what the oracle finishes in seconds: size-independent properties instead --
    Parseval, sampled bins against a direct O(N) DFT (exact integer phase reduction), and the round trip."""
    import torch

    n = 1 << k
    tdt = torch.float64 if dt == "f64" else torch.float32
    Planner = gpu.PlannerDit64 if dt == "f64" else gpu.PlannerDit32
    fwd = gpu.fft_64_dit_with_planner if dt == "f64" else gpu.fft_32_dit_with_planner
    re = torch.empty(n, dtype=tdt, device="cuda")
    im = torch.empty_like(re)
    gpu.fill_uniform(re, im, n, seed=0xCAFE, first_id=k)
    re0, im0 = re.clone(), im.clone()
    planner = Planner(n)
    fwd(re, im, gpu.Direction.Forward, planner)
    e_in = float((re0.double() ** 2 + im0.double() ** 2).sum())
    e_out = float((re.double() ** 2 + im.double() ** 2).sum())
    assert abs(e_out / (n * e_in) - 1.0) < tol.parseval_gate(dt, k), planner.describe()
    j = torch.arange(n, dtype=torch.int64, device="cuda")
    norm = np.sqrt(n * e_in)
    for kk in (0, 1, 777, n // 2 + 3, n - 1):
        ang = ((j * kk) % n).to(torch.float64) * (-2.0 * np.pi / n)
        c, s = torch.cos(ang), torch.sin(ang)
        xr = float((re0.double() * c - im0.double() * s).sum())
        xi = float((re0.double() * s + im0.double() * c).sum())
        gate = tol.bin_gate(dt, k) * norm / np.sqrt(n)   # per bin, relative to the rms bin = ||X|| / sqrt(N)
        assert abs(float(re[kk]) - xr) < gate and abs(float(im[kk]) - xi) < gate, (kk, planner.describe())
        del ang, c, s
    del j
    fwd(re, im, gpu.Direction.Reverse, planner)
    lim = 1e-10 if dt == "f64" else 2e-5
    assert float((re - re0).abs().max()) < lim and float((im - im0).abs().max()) < lim


def test_strided_batches_and_untouched_gaps(gpu, oracle):
    """Batched device path with dist > n: every transform right, the gaps between transforms untouched; both
    plan families (small-LDS kernel at 2^9, tile passes at 2^13)."""
    import torch

    for k, batch, gap in ((9, 5, 7), (13, 3, 64), (16, 2, 8)):
        n = 1 << k
        dist = n + gap
        total = (batch - 1) * dist + n
        re = torch.full((total,), 123.0, dtype=torch.float64, device="cuda")
        im = torch.full((total,), -321.0, dtype=torch.float64, device="cuda")
        for b in range(batch):
            r, m = oracle.fill(n, np.float64, transform_id=50 + b)
            re[b * dist:b * dist + n] = torch.from_numpy(r).cuda()
            im[b * dist:b * dist + n] = torch.from_numpy(m).cuda()
        planner = gpu.PlannerDit64(n)
        gpu.fft_dit_batched(re, im, n, gpu.Direction.Forward, planner, dist=dist)
        hre, him = re.cpu().numpy(), im.cpu().numpy()
        for b in range(batch):
            r, m = oracle.fill(n, np.float64, transform_id=50 + b)
            oracle.fft_64_dit(r, m, oracle.FORWARD)
            tol.check("batched_dist_vs_oracle", "f64", k, hre[b * dist:b * dist + n], him[b * dist:b * dist + n], r, m)
            if b + 1 < batch:
                assert np.all(hre[b * dist + n:(b + 1) * dist] == 123.0) and np.all(him[b * dist + n:(b + 1) * dist] == -321.0)


def test_batched_r2c_c2r(gpu, oracle):
    import torch

    for k, batch in ((8, 9), (14, 5)):
        n = 1 << k
        x = torch.empty(batch * n, dtype=torch.float32, device="cuda")
        gpu.fill_uniform(x, None, n, seed=0xCAFE, first_id=300)
        ore = torch.empty(batch * (n // 2 + 1), dtype=torch.float32, device="cuda")
        oim = torch.empty_like(ore)
        planner = gpu.PlannerR2c32(n)
        gpu.r2c_fft_batched(x, ore, oim, planner, batch)
        hre, him = ore.cpu().numpy(), oim.cpu().numpy()
        for b in range(batch):
            hx, _ = oracle.fill(n, np.float32, seed=0xCAFE, transform_id=300 + b)
            rr, ri = np.zeros(n // 2 + 1, np.float32), np.zeros(n // 2 + 1, np.float32)
            oracle.r2c_fft_f32(hx, rr, ri)
            sl = slice(b * (n // 2 + 1), (b + 1) * (n // 2 + 1))
            tol.check("batched_r2c_f32_vs_oracle", "f32", k, hre[sl], him[sl], rr.astype(np.float64), ri.astype(np.float64), against="oracle")
            ind = np.fft.rfft(hx.astype(np.float64))
            tol.check("batched_r2c_f32_vs_rfft", "f32", k, hre[sl], him[sl], ind.real, ind.imag)
        back = torch.empty_like(x)
        gpu.c2r_fft_batched(ore, oim, back, planner, batch)
        assert float((back - x).abs().max()) < 5 * tol.ROUNDTRIP_ABS["f32"]


@pytest.mark.parametrize("k,batch,dt", [(20, 64, "f32"), (18, 256, "f32"), (20, 64, "f64"), (24, 4, "f64")])
def test_batched_r2c_c2r_throughput_plans(gpu, oracle, k, batch, dt):
    """Enough real transforms in flight that the inner complex FFT runs its throughput plan (wide tiles with the
    fused deinterleaving load / interleaving store): R2C against the oracle, C2R back to the input."""
    import torch

    n = 1 << k
    tdt, ndt = (torch.float64, np.float64) if dt == "f64" else (torch.float32, np.float32)
    planner = (gpu.PlannerR2c64 if dt == "f64" else gpu.PlannerR2c32)(n)
    x = torch.empty(batch * n, dtype=tdt, device="cuda")
    gpu.fill_uniform(x, None, n, seed=0xF00D, first_id=11)
    ore = torch.empty(batch * (n // 2 + 1), dtype=tdt, device="cuda")
    oim = torch.empty_like(ore)
    gpu.r2c_fft_batched(x, ore, oim, planner, batch)
    ofn = oracle.r2c_fft_f64 if dt == "f64" else oracle.r2c_fft_f32
    for b in (0, batch // 2, batch - 1):
        hx, _ = oracle.fill(n, ndt, seed=0xF00D, transform_id=11 + b)
        rr, ri = np.zeros(n // 2 + 1, ndt), np.zeros(n // 2 + 1, ndt)
        ofn(hx, rr, ri)
        sl = slice(b * (n // 2 + 1), (b + 1) * (n // 2 + 1))
        g_re, g_im = ore[sl].cpu().numpy(), oim[sl].cpu().numpy()
        tol.check("batched_r2c_tp_vs_oracle", dt, k, g_re, g_im, rr.astype(np.float64), ri.astype(np.float64), against="oracle_real")
        ind = np.fft.rfft(hx.astype(np.float64))   # ... and the tight gate against an independent real FFT
        tol.check("batched_r2c_tp_vs_rfft", dt, k, g_re, g_im, ind.real, ind.imag)
    back = torch.empty_like(x)
    gpu.c2r_fft_batched(ore, oim, back, planner, batch)
    assert float((back - x).abs().max()) < (1e-12 if dt == "f64" else 10 * tol.ROUNDTRIP_ABS["f32"])


def test_one_planner_shared_by_host_threads(gpu, oracle):
    """planner.rs:38-39: a planner is an immutable value any number of callers may borrow.  Here it owns device
    scratch, so the library serialises the calls; four host threads hammer one C2C and one R2C planner with
    different data (ctypes releases the GIL during the calls) and every result must still be right."""
    import threading

    n = 1 << 16
    planner = gpu.PlannerDit64(n)
    rplanner = gpu.PlannerR2c32(n)
    errors = []

    def worker(tid):
        try:
            for it in range(12):
                re, im = oracle.fill(n, np.float64, transform_id=1000 * tid + it)
                a, b = re.copy(), im.copy()
                gpu.fft_64_dit_with_planner(a, b, gpu.Direction.Forward, planner)
                oracle.fft_64_dit(re, im, oracle.FORWARD)
                if rel_l2(a, b, re, im) > tol.f64_rel(16):
                    errors.append(("c2c", tid, it))
                x, _ = oracle.fill(n, np.float32, transform_id=5000 * tid + it)
                ore, oim = np.zeros(n // 2 + 1, np.float32), np.zeros(n // 2 + 1, np.float32)
                gpu.r2c_fft_f32_with_planner(x, ore, oim, rplanner)
                rr, ri = np.zeros(n // 2 + 1, np.float32), np.zeros(n // 2 + 1, np.float32)
                oracle.r2c_fft_f32(x, rr, ri)
                if rel_l2(ore, oim, rr, ri) > tol.F32_REL_VS_ORACLE:
                    errors.append(("r2c", tid, it))
        except Exception as e:  # noqa: BLE001 -- surfaced below
            errors.append(("exception", tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_small_transform_batches_ragged_and_strided(gpu, oracle, dt):
    """The one-pass kernel for N <= 8192 (row_fft.hpp): batches that do not fill the last workgroup tile,
    transforms `dist` apart, forward and inverse; first / middle / last transform against the oracle, gaps untouched."""
    import torch

    tdt, ndt = (torch.float64, np.float64) if dt == "f64" else (torch.float32, np.float32)
    ofn = oracle.fft_64_dit if dt == "f64" else oracle.fft_32_dit
    for k in range(0, 14):
        n = 1 << k
        per_tile = max(1, 4096 // n) if k >= 6 else 256
        batch = 2 * per_tile + per_tile // 2 + 3
        gap = 5
        dist = n + gap
        total = (batch - 1) * dist + n
        planner = (gpu.PlannerDit64 if dt == "f64" else gpu.PlannerDit32)(n)
        for direction, odir in ((gpu.Direction.Forward, oracle.FORWARD), (gpu.Direction.Reverse, oracle.REVERSE)):
            h_re = np.full(total, 9.0, ndt)
            h_im = np.full(total, -9.0, ndt)
            picks = sorted({0, 1 % batch, per_tile - 1, per_tile, batch // 2, batch - 1})
            for b in range(batch):
                r, m = oracle.fill(n, ndt, seed=0xABCD, transform_id=b)
                h_re[b * dist:b * dist + n], h_im[b * dist:b * dist + n] = r, m
            re, im = torch.from_numpy(h_re).cuda(), torch.from_numpy(h_im).cuda()
            gpu.fft_dit_batched(re, im, n, direction, planner, dist=dist)
            g_re, g_im = re.cpu().numpy(), im.cpu().numpy()
            for b in picks:
                r, m = oracle.fill(n, ndt, seed=0xABCD, transform_id=b)
                ofn(r, m, odir)
                sl = slice(b * dist, b * dist + n)
                tol.check("small_ragged_vs_oracle", dt, k, g_re[sl], g_im[sl], r.astype(np.float64), m.astype(np.float64), against="oracle")
            mask = np.ones(total, bool)
            for b in range(batch):
                mask[b * dist:b * dist + n] = False
            assert np.all(g_re[mask] == 9.0) and np.all(g_im[mask] == -9.0), k


# ---------------------------------------------------------------- one transform over several ranks (f-3)
def test_twiddle_grid_kernel(gpu):
    import torch

    n = 1 << 24
    for dt, Grid, tol in ((torch.float64, gpu.TwiddleGrid64, 2e-15), (torch.float32, gpu.TwiddleGrid32, 4e-7)):
        rows, cols, row0, col0 = 37, 1000, (1 << 11) + 5, 3
        re = torch.rand(rows * cols, dtype=dt, device="cuda") - 0.5
        im = torch.rand(rows * cols, dtype=dt, device="cuda") - 0.5
        x, y = re.double().clone(), im.double().clone()
        Grid(n).apply(re, im, rows, cols, row0=row0, col0=col0)
        r = torch.arange(row0, row0 + rows, dtype=torch.int64, device="cuda").view(rows, 1)
        c = torch.arange(col0, col0 + cols, dtype=torch.int64, device="cuda").view(1, cols)
        ang = ((r * c) % n).to(torch.float64) * (-2.0 * np.pi / n)
        wr, wi = torch.cos(ang).view(-1), torch.sin(ang).view(-1)
        assert float((re.double() - (x * wr - y * wi)).abs().max()) < tol
        assert float((im.double() - (x * wi + y * wr)).abs().max()) < tol


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_four_step_on_one_rank_matches_the_library(gpu, dt):
    """world = 1: the distributed path's local stages (two batched FFTs + the twiddle grid) against the
    library's own transform of the same data."""
    import torch

    from phastft_amd.distributed import gpu_transform

    n = 1 << 21
    tdt = torch.float64 if dt == "f64" else torch.float32
    re = torch.empty(n, dtype=tdt, device="cuda")
    im = torch.empty_like(re)
    gpu.fill_uniform(re, im, n, seed=0x5EED)
    a, b = re.clone(), im.clone()
    gpu_transform(n, 0, 1, None, dt).run(a, b)
    (gpu.fft_64_dit if dt == "f64" else gpu.fft_32_dit)(re, im, gpu.Direction.Forward)
    scale = float((re.double() ** 2 + im.double() ** 2).sum().sqrt())
    err = float(((a.double() - re.double()) ** 2 + (b.double() - im.double()) ** 2).sum().sqrt()) / scale
    assert err < (1e-14 if dt == "f64" else 1e-6), err


def _two_rank_worker(rank, world, port, log_n, out_dir):
    import os

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import phastft_amd as P
    from phastft_amd.distributed import gpu_transform

    torch.cuda.set_device(0)  # dry run: both ranks share the one GPU of the box; blocks travel through host memory
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1 << log_n
    full_re = torch.empty(n, dtype=torch.float64, device="cuda")
    full_im = torch.empty_like(full_re)
    P.fill_uniform(full_re, full_im, n, seed=0xD157)
    lo, hi = rank * n // world, (rank + 1) * n // world
    re, im = full_re[lo:hi].clone(), full_im[lo:hi].clone()
    t = gpu_transform(n, rank, world, dist, "f64")
    t.run(re, im)
    P.fft_64_dit(full_re, full_im, P.Direction.Forward)  # the whole transform on one GPU, for comparison
    scale = float((full_re ** 2 + full_im ** 2).sum().sqrt())
    err = float(((re - full_re[lo:hi]) ** 2 + (im - full_im[lo:hi]) ** 2).sum().sqrt()) / scale
    t.run(re, im, reverse=True)
    P.fill_uniform(full_re, full_im, n, seed=0xD157)
    back = max(float((re - full_re[lo:hi]).abs().max()), float((im - full_im[lo:hi]).abs().max()))
    with open(os.path.join(out_dir, f"err{rank}.txt"), "w") as f:
        f.write(f"{err} {back}")
    dist.barrier()
    dist.destroy_process_group()


def test_one_transform_over_two_ranks_sharing_the_gpu(gpu, tmp_path):
    """The multi-GPU single-transform path end to end with the real kernels: two processes, one GPU, gloo carrying
    the three all-to-alls through host memory (RCCL itself needs more than one GPU)."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_two_rank_worker, args=(2, port, 22, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        err, back = (float(v) for v in open(tmp_path / f"err{r}.txt").read().split())
        assert err < 1e-14 and back < 1e-12, (r, err, back)


def test_wave_tiles_all_passes_batched_inverse_and_interleaved(gpu, oracle, static_rules):
    """wave_fft.hpp (one wave per 64 x 16 tile, cross-lane swaps): N = 2^18 as three wave-tile passes -- first pass
    (transposing through the wave-private buffer), pre-twiddle passes, batched with a ragged tile count per workgroup,
    the inverse (1/N in the last store), and the interleaved first-pass load / last-pass store."""
    import torch

    n = 1 << 18
    planner = gpu.PlannerDit64(n)
    planner.set_plan((6, 6, 6), 10, 4 | 0x10)
    assert planner.describe().count(" w16 ") == 3, planner.describe()
    for batch in (1, 3):
        re = torch.empty(batch * n, dtype=torch.float64, device="cuda")
        im = torch.empty_like(re)
        gpu.fill_uniform(re, im, n, seed=0x77, first_id=9)
        gpu.fft_dit_batched(re, im, n, gpu.Direction.Forward, planner)
        for b in range(batch):
            r, m = oracle.fill(n, np.float64, seed=0x77, transform_id=9 + b)
            oracle.fft_64_dit(r, m, oracle.FORWARD)
            tol.check("wave_quad_batch_vs_oracle", "f64", n.bit_length() - 1, re[b * n:(b + 1) * n].cpu().numpy(), im[b * n:(b + 1) * n].cpu().numpy(), r, m)
        gpu.fft_dit_batched(re, im, n, gpu.Direction.Reverse, planner)
        ref_re, ref_im = torch.empty_like(re), torch.empty_like(im)
        gpu.fill_uniform(ref_re, ref_im, n, seed=0x77, first_id=9)
        assert float((re - ref_re).abs().max()) < 1e-12 and float((im - ref_im).abs().max()) < 1e-12
    r, m = oracle.fill(n, np.float64, seed=0x78, transform_id=1)
    z = np.empty(n, np.complex128)
    z.real, z.imag = r, m
    d = torch.from_numpy(z.copy()).cuda()
    gpu.fft_64_interleaved_with_planner(d, gpu.Direction.Forward, planner)
    oracle.fft_64_dit(r, m, oracle.FORWARD)
    h = d.cpu().numpy()
    tol.check("wave_quad_interleaved_vs_oracle", "f64", n.bit_length() - 1, h.real.copy(), h.imag.copy(), r, m)
    # the plans for ONE transform of 2^14, 2^15 and 2^19 .. 2^21 points ARE wave- / quad-tile plans (round 4: from 2^22 on the
    # cold-ring sweep put generic tiles back, plan.hpp: single_plan)
    for k in (14, 15, 19, 20, 21):
        lat = gpu.PlannerDit64(1 << k).describe().split("single=")[1]
        assert " w16 " in lat or " q16 " in lat, (k, lat)
    for k in (22, 23, 24, 25, 28):
        lat = gpu.PlannerDit64(1 << k).describe().split("single=")[1]
        assert " w16 " not in lat and " q16 " not in lat and lat.startswith("3p["), (k, lat)
    assert " q16 " in gpu.PlannerDit64(1 << 20).describe().split("single=")[1]
