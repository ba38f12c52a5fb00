"""Register budget of the compiled kernels (phastft_amd/lib/kernel_resources.json, written by phastft_amd/build.py from
hipcc's kernel-resource-usage remarks).  A kernel that drifts into scratch memory -- a register array the optimiser
could not keep in registers, or plain spills -- keeps passing every parity test while paying for it in HBM traffic
(round 2: 80 bytes per lane of scratch in the wave/quad kernels showed up as +10 % FETCH_SIZE/WRITE_SIZE), so the
build records the numbers and this test pins them."""
import json
import os

import pytest

import phastft_amd.build as B

# kernels known to spill, with the bytes per lane they are allowed.  Round 3: NO kernel of any default plan is here any
# more -- the two widest pass kernels (1024 x 16 f64 and 1024 x 32 f32 at 32 points per thread, the dominant kernels of
# BASELINE configs[4] and of its f32 twin) spilled 10 and 19 VGPRs in round 2 and use no scratch now (their own
# translation units, tile_f64_bc_wide.hip / tile_f32_bc_wide.hip).
KNOWN_SCRATCH = {}  # round 4: the one kernel that used scratch (a bit-reversal tuning variant) is gone -- keep it empty
WIDEST = ("_ZN5phast15tile_fft_kernelIdLi10ELi4ELi5ELb1ELb0ELb1EEEvNS_8TileArgsE",
          "_ZN5phast15tile_fft_kernelIfLi10ELi5ELi5ELb1ELb0ELb1EEEvNS_8TileArgsE")


@pytest.fixture(scope="module")
def resources():
    B.build()
    assert os.path.exists(B.RESOURCES), "build.py did not write kernel_resources.json"
    return json.load(open(B.RESOURCES))


def test_every_unit_reported(resources):
    names = " ".join(resources)
    for kernel in ("tile_fft_kernel", "wave_fft_kernel", "quad_fft_kernel", "row_fft", "bitrev_persistent2_kernel",
                   "untangle_kernel", "c2r_preprocess", "twiddle_grid", "fill_kernel"):
        assert kernel in names, kernel
    assert len(resources) > 100


def test_no_kernel_uses_scratch_memory_unannounced(resources):
    bad = {k: v["scratch"] for k, v in resources.items() if v.get("scratch", 0) > KNOWN_SCRATCH.get(k, 0)}
    assert not bad, bad
    for k in KNOWN_SCRATCH:
        assert k in resources, f"{k} no longer exists: drop it from KNOWN_SCRATCH"


def test_wave_and_quad_kernels_keep_four_waves_per_simd(resources):
    """the headline kernels: the one-wave 64-row tiles and the four-wave 256-row tile (f64: 16 columns; f32, round 6: 32 columns
    as float2 pairs), no spills, <= 128 VGPRs"""
    seen = 0
    for k, v in resources.items():
        if "wave_fft_kernel" in k or "quad_fft_kernel" in k:
            seen += 1
            assert v["scratch"] == 0 and v["vgpr_spill"] == 0, (k, v)
            if "kernelIf" in k:   # f32 (round 6): built with max-ilp (build.py: UNIT_FLAGS) -- step twiddles prefetched, more registers
                assert v["vgprs"] <= 256 and v["occupancy"] >= 2, (k, v)
            else:
                assert v["vgprs"] <= 128 and v["occupancy"] >= 4, (k, v)
    assert seen >= 8   # wave: {f64, f32} x {first pass, later pass}; quad: {f64, f32} x {persistent, one tile per workgroup}


def test_widest_pass_kernels_do_not_spill(resources):
    """0 bytes of scratch and no spilled VGPRs in the dominant kernels of configs[4] (f64) and of
    the batched f32 2^20 transforms; every tile / wave / quad / row kernel of every plan is scratch-free."""
    for k in WIDEST:
        assert k in resources, k
        assert resources[k]["scratch"] == 0 and resources[k]["vgpr_spill"] == 0, (k, resources[k])
    for k, v in resources.items():
        if any(t in k for t in ("tile_fft_kernel", "wave_fft_kernel", "quad_fft_kernel", "row_fft_kernel")):
            assert v["scratch"] == 0, (k, v)
