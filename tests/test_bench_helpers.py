"""bench.py's host-side helpers (no GPU): the parsing of planner.describe() that decides which plan a call runs, how the
plan is named on the JSON line and which profiled kernels its passes map to (profiles/traffic_latest.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

PLAN_20 = ("n=2^20 throughput=2p[1024x16A p32 lds=136192 wg/cu=1][1024x16 p32 lds=142336 wg/cu=1] "
           "mid=2p[1024x8A p16 lds=70656 wg/cu=2][1024x8 p16 lds=76800 wg/cu=2] "
           "latency=3p[64x64A p8 lds=67584 wg/cu=2][256x16 p8 lds=76288 wg/cu=2][64x64 p8 lds=72704 wg/cu=2] "
           "single=3p[64x16A w16 lds=67072 wg/cu=2][256x16 q16 lds=69120 wg/cu=2][64x16 w16 lds=6656 wg/cu=4]")
PLAN_R2C = ("n=2^23 throughput=3p[256x32A p16 lds=66304 wg/cu=2][256x32 p16 lds=67584 wg/cu=2][128x64 p16 lds=72192 wg/cu=2] "
            "latency=3p[256x16A p8 lds=37376 wg/cu=4][256x16 p8 lds=38912 wg/cu=4][128x32 p8 lds=39424 wg/cu=4] "
            "single=3p[256x16A p16 lds=35328 wg/cu=4][256x16 p16 lds=36864 wg/cu=4][128x32 p16 lds=39936 wg/cu=4]")


def test_plan_kind_follows_the_library():
    n = 1 << 20
    assert bench.plan_kind(None, n, 1, PLAN_20) == "single"
    assert bench.plan_kind(None, n, 2, PLAN_20) == "single"
    assert bench.plan_kind(None, n, 3, PLAN_20) == "mid"
    assert bench.plan_kind(None, n, 15, PLAN_20) == "mid"
    assert bench.plan_kind(None, n, 1, PLAN_20.split(" single=")[0]) == "latency"
    assert bench.plan_kind(None, n, 16, PLAN_20) == "throughput"      # 2^24 points in flight, 16384-point tiles
    assert bench.plan_kind(None, n, 1024, PLAN_20) == "throughput"
    assert bench.plan_kind(None, 1 << 23, 1, PLAN_R2C, "f32") == "single"
    assert bench.plan_kind(None, 1 << 23, 3, PLAN_R2C, "f32") == "latency"   # f32: the crossover sits one octave higher
    assert bench.plan_kind(None, 1 << 23, 4, PLAN_R2C, "f32") == "throughput"
    assert bench.plan_kind(None, 1 << 10, 64, "n=2^10 one pass (whole transforms on chip)") == "one-pass"


def test_plan_lists_and_kernel_names():
    lat = bench.plan_of(PLAN_20, "single")
    assert lat.startswith("[64x16A w16") and lat.count("[") == 3
    assert bench.kernel_tags(lat) == ["wave_fft_kernel<double, false, true>", "quad_fft_kernel<double>",
                                      "wave_fft_kernel<double, true, false>"]
    thr = bench.plan_of(PLAN_20, "throughput")
    assert bench.kernel_tags(thr) == ["tile_fft_kernel<double, 10, 4, 5, false, true,",
                                      "tile_fft_kernel<double, 10, 4, 5, true, false,"]
    assert bench.kernel_tags(bench.plan_of(PLAN_R2C, "single"), "float")[2] == "tile_fft_kernel<float, 7, 5, 4, true, false,"


def test_committed_traffic_profile_matches_the_default_plans():
    """profiles/traffic_latest.json must belong to the plans the library runs today, or bench.py reports traffic = null."""
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    for key in ("single_2p20", "single_2p26", "r2c_f32_2p24", "batch_2p20"):
        assert key in t and t[key]["kernels"], key
    tr = bench.load_profiled_traffic(1, 1, 3, bench.kernel_tags(bench.plan_of(PLAN_20, "single")))
    assert tr is not None and "quad_fft_kernel" in tr["traffic_kernel"]
    assert 1.0 <= tr["traffic"] / 33554432 < 1.02       # HBM bytes ~ algorithmic bytes
    tb = bench.load_profiled_traffic(8, 0, 2, bench.kernel_tags(bench.plan_of(PLAN_20, "throughput")))
    assert tb is not None and 1.0 <= tb["traffic"] / (1024 * 33554432) < 1.02  # profiled per 1024-transform launch (round 3)


def test_roofline_arithmetic():
    roof, dom = bench.roofline_of([0.008, 0.010, 0.008], 33554432, plan_used="x")
    assert dom == 1 and abs(roof["achieved"] - 33554432 / 0.010e-3 / 1e9) < 1e-6
    assert abs(roof["frac"] - roof["achieved"] / 8000.0) < 1e-12
    assert abs(roof["transform_frac"] - 33554432 / 0.026e-3 / 1e9 / 8000.0) < 1e-9 and roof["passes"] == 3


# ---------------------------------------------------------------- round 4: `bench.py --gpus N` cannot mis-measure
def _run_bench(argv, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE")):
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, env=env,
                          timeout=300, cwd=ROOT)


def test_bench_refuses_gpus_it_does_not_have():
    """A plain `python bench.py --gpus 8` on a box with fewer GPUs (none here) must exit non-zero with the reason and
    print NO JSON line -- round 3 printed a warning and measured one GPU (missing #1)."""
    import torch

    if torch.cuda.device_count() >= 8:
        import pytest

        pytest.skip("this box really has 8 GPUs")
    r = _run_bench(["--gpus", "8", "--steps", "1"])
    assert r.returncode != 0
    assert "GPU(s) visible" in r.stderr and "nothing measured" in r.stderr
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_bench_refuses_world_size_mismatch():
    """--gpus must equal the number of ranks the launcher started: never a 2-rank number labelled n_gpus 1 or vice versa"""
    r = _run_bench(["--gpus", "1", "--steps", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())
    r = _run_bench(["--gpus", "2", "--same-gpu", "--steps", "1"])          # RCCL cannot put two ranks on one device
    assert r.returncode != 0 and "gloo" in r.stderr


def test_bench_self_launch_command_is_the_drivers():
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "3"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-5].endswith("bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
