"""bench.py's host-side helpers (no GPU): how the plan a call runs is named on the JSON line and which profiled kernels its
passes map to (profiles/traffic_latest.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

SINGLE_20 = "[64x16A w16][256x16 q16][64x16 w16]"          # what Planner::describe_call answers for one f64 2^20 transform
THROUGHPUT_20 = "[1024x16A p32][1024x16 p32]"
R2C_SINGLE_23 = "[256x16A p16][256x16 p16][128x32 p16]"


class _FakePlanner:
    """stands in for phastft_amd.PlannerDit64: bench.py labels its numbers with the LIBRARY's answer (describe_call ->
    Planner::choose), not with a copy of the plan rules (round 5)"""

    def __init__(self, text):
        self.text, self.asked = text, None

    def describe_call(self, batch, kind):
        self.asked = (batch, kind)
        return self.text


def test_plan_label_is_the_librarys_answer():
    pl = _FakePlanner("single " + SINGLE_20)
    assert bench.plan_used(pl, 1) == ("single", SINGLE_20) and pl.asked == (1, 0)
    pl = _FakePlanner("tuned [128x32A p8][128x32 p8][64x64 p8] untangle-fused")
    which, lst = bench.plan_used(pl, 16, 2)
    assert which == "tuned" and lst.endswith("untangle-fused") and pl.asked == (16, 2)
    assert bench.plan_used(_FakePlanner("one-pass"), 64) == ("one-pass", "")


def test_plan_lists_and_kernel_names():
    assert bench.kernel_tags(SINGLE_20) == ["wave_fft_kernel<double, false, true>", "quad_fft_kernel<double",
                                            "wave_fft_kernel<double, true, false>"]
    assert bench.kernel_tags(THROUGHPUT_20) == ["tile_fft_kernel<double, 10, 4, 5, false, true,",
                                                "tile_fft_kernel<double, 10, 4, 5, true, false,"]
    assert bench.kernel_tags(R2C_SINGLE_23, "float")[2] == "tile_fft_kernel<float, 7, 5, 4, true, false,"


def test_committed_traffic_profile_matches_the_default_plans():
    """profiles/traffic_latest.json must belong to the plans the library runs today, or bench.py reports traffic = null."""
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    for key in ("single_2p20", "single_2p26", "r2c_f32_2p24", "batch_2p20"):
        assert key in t and t[key]["kernels"], key
    tr = bench.load_profiled_traffic(1, 1, 3, bench.kernel_tags(SINGLE_20))
    assert tr is not None and "quad_fft_kernel" in tr["traffic_kernel"]
    assert 1.0 <= tr["traffic"] / 33554432 < 1.02       # HBM bytes ~ algorithmic bytes
    tb = bench.load_profiled_traffic(8, 0, 2, bench.kernel_tags(THROUGHPUT_20))
    assert tb is not None and 1.0 <= tb["traffic"] / (1024 * 33554432) < 1.02  # profiled per 1024-transform launch (round 3)


def test_roofline_arithmetic():
    roof, dom = bench.roofline_of([0.008, 0.010, 0.008], 33554432, plan_used="x")
    assert dom == 1 and abs(roof["achieved"] - 33554432 / 0.010e-3 / 1e9) < 1e-6
    assert abs(roof["frac"] - roof["achieved"] / 8000.0) < 1e-12
    assert roof["frac_dominant_pass"] == roof["frac"]
    assert abs(roof["frac_transform"] - 33554432 / 0.026e-3 / 1e9 / 8000.0) < 1e-9 and roof["passes"] == 3


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


# ---------------------------------------------------------------- the N > 1 line's schema
def validate_multi_gpu_line(line: dict):
    """tests/golden/multi_gpu_line.schema.json: the keys both hosts of BASELINE configs[4] print.  Returns a list of problems."""
    import json

    schema = json.load(open(os.path.join(ROOT, "tests", "golden", "multi_gpu_line.schema.json")))
    ok_type = {"string": lambda v: isinstance(v, str), "number": lambda v: isinstance(v, (int, float)) and not isinstance(v, bool),
               "integer": lambda v: isinstance(v, int) and not isinstance(v, bool), "boolean": lambda v: isinstance(v, bool),
               "object": lambda v: isinstance(v, dict), "null-or-number": lambda v: v is None or isinstance(v, (int, float))}
    bad = []
    for k, t in schema["top"]["required"].items():
        if k not in line:
            bad.append(f"missing {k}")
        elif not ok_type[t](line[k]):
            bad.append(f"{k}: not a {t}")
    for k, v in schema["top"]["fixed"].items():
        if line.get(k) != v:
            bad.append(f"{k} != {v!r}")
    cfg = line.get("config", {})
    for k, t in schema["config"]["required"].items():
        if k not in cfg:
            bad.append(f"missing config.{k}")
        elif not ok_type[t](cfg[k]):
            bad.append(f"config.{k}: not a {t}")
    for k, t in schema["config"]["optional"].items():
        if k in cfg and not ok_type[t](cfg[k]):
            bad.append(f"config.{k}: not a {t}")
    if not bad:
        if cfg["ranks_seen"] != line["n_gpus"]:
            bad.append("config.ranks_seen != n_gpus")
        if cfg["rank_ms_min"] > cfg["rank_ms_max"]:
            bad.append("rank_ms_min > rank_ms_max")
        if abs(cfg["rank_ms_max"] - line["ms_per_step"]) > 0.02 * line["ms_per_step"] + 0.05:
            bad.append("rank_ms_max is not the step time")
    return bad


def test_multi_gpu_schema_and_both_hosts_name_its_keys():
    """No GPU here: the schema loads, a well-formed line passes, broken ones are caught -- and the SOURCES of the two hosts
    (bench.py's N > 1 branch, tests/cpp/shard_host.cpp's printf) carry every required key of `config` (the run itself is
    tests/test_gpu_parity_r5.py::test_both_hosts_print_the_same_multi_gpu_schema on a GPU box)."""
    import json

    schema = json.load(open(os.path.join(ROOT, "tests", "golden", "multi_gpu_line.schema.json")))
    good = {"metric": "GSamples/s f64 forward FFT N=2^20", "value": 660.0, "unit": "GSamples/s", "n_gpus": 8, "steps": 10, "warmup": 2,
            "ms_per_step": 13.0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "8192 ...", "plan_used": "throughput", "digest_gather": "all_gather ...", "digest_ok": True,
                       "rank_ms_min": 12.7, "rank_ms_max": 13.0, "ranks_seen": 8, "backend": "nccl", "shard": 1024, "rccl_version": "2.26.6",
                       "xgmi_links": 28}}
    assert validate_multi_gpu_line(good) == []
    assert validate_multi_gpu_line(dict(good, n_gpus=4)) == ["config.ranks_seen != n_gpus"]
    broken = json.loads(json.dumps(good))
    del broken["config"]["rank_ms_min"]
    broken["config"]["shard"] = "1024"
    assert validate_multi_gpu_line(broken) == ["missing config.rank_ms_min", "config.shard: not a integer"]
    bench_src = open(os.path.join(ROOT, "bench.py")).read()
    host_src = open(os.path.join(ROOT, "tests", "cpp", "shard_host.cpp")).read()
    for k in schema["config"]["required"]:
        assert f'"{k}"' in bench_src, f"bench.py does not print config.{k}"
        assert f'\\"{k}\\"' in host_src, f"shard_host.cpp does not print config.{k}"
