"""Loader for libphastft_hip.so (the C ABI of include/phastft_hip.h).

The HIP library IS the product: there is no CPU path behind this module.  If the shared object is
missing or a call fails, the caller gets an exception -- never a silent fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PHASTFT_HIP_LIB") or os.path.join(_HERE, "lib", "libphastft_hip.so")

# every symbol include/phastft_hip.h declares (tests/test_abi.py checks this list against the header)
SYMBOLS = """
phast_strerror phast_last_hip_error phast_device_info phast_options_default phast_options_guess
phast_planner_dit64_new phast_planner_dit64_with_mode phast_planner_dit64_free
phast_planner_dit32_new phast_planner_dit32_with_mode phast_planner_dit32_free
phast_planner_dit64_device_bytes phast_planner_dit32_device_bytes
phast_planner_dit64_describe phast_planner_dit32_describe
phast_planner_dit64_reserve_batch phast_planner_dit32_reserve_batch
phast_planner_r2c64_new phast_planner_r2c64_free phast_planner_r2c32_new phast_planner_r2c32_free
phast_fft_64_dit phast_fft_32_dit phast_fft_64_dit_with_planner phast_fft_32_dit_with_planner
phast_fft_64_dit_with_planner_and_opts phast_fft_32_dit_with_planner_and_opts
phast_fft_64_dit_dev phast_fft_32_dit_dev phast_fft_64_dit_many_dev phast_fft_32_dit_many_dev phast_fft_64_dit_strided_dev phast_fft_32_dit_strided_dev phast_fft_64_dit_strided_tw_dev phast_fft_32_dit_strided_tw_dev
phast_fft_64_interleaved phast_fft_32_interleaved phast_fft_64_interleaved_with_planner
phast_fft_32_interleaved_with_planner phast_fft_64_interleaved_with_planner_and_opts
phast_fft_32_interleaved_with_planner_and_opts phast_fft_64_interleaved_dev phast_fft_32_interleaved_dev
phast_bit_rev_f64 phast_bit_rev_f32 phast_bit_rev_f64_dev phast_bit_rev_f32_dev
phast_deinterleave_f64 phast_deinterleave_f32 phast_deinterleave_f64_dev phast_deinterleave_f32_dev
phast_combine_re_im_f64 phast_combine_re_im_f32 phast_combine_re_im_f64_dev phast_combine_re_im_f32_dev
phast_r2c_fft_f64 phast_r2c_fft_f32 phast_r2c_fft_f64_with_planner phast_r2c_fft_f32_with_planner
phast_r2c_fft_f64_dev phast_r2c_fft_f32_dev
phast_c2r_fft_f64 phast_c2r_fft_f32 phast_c2r_fft_f64_with_planner phast_c2r_fft_f32_with_planner
phast_c2r_fft_f64_with_planner_and_scratch phast_c2r_fft_f32_with_planner_and_scratch
phast_c2r_fft_f64_dev phast_c2r_fft_f32_dev
phast_fill_f64_dev phast_fill_f32_dev phast_digest_f64_dev phast_digest_f32_dev phast_hip_graph_upload phast_stream_probe_dev
phast_planner_dit64_set_plan phast_planner_dit32_set_plan
phast_planner_dit64_time_passes phast_planner_dit32_time_passes phast_debug_set_wg_per_cu phast_debug_set_trace
phast_debug_set_guard_bytes phast_planner_dit64_debug_check_guards phast_planner_dit32_debug_check_guards
phast_planner_r2c64_time_passes phast_planner_r2c32_time_passes phast_planner_r2c64_time_c2r_passes phast_planner_r2c32_time_c2r_passes phast_planner_r2c64_describe phast_planner_r2c32_describe phast_planner_r2c64_set_inner_plan phast_planner_r2c32_set_inner_plan
phast_twiddle_grid64_new phast_twiddle_grid32_new phast_twiddle_grid64_free phast_twiddle_grid32_free
phast_twiddle_grid64_apply_dev phast_twiddle_grid32_apply_dev
phast_planner_dit64_release_graph_workspaces phast_planner_dit32_release_graph_workspaces
phast_planner_dit64_tune phast_planner_dit32_tune phast_planner_r2c64_tune phast_planner_r2c32_tune
phast_planner_r2c64_with_mode phast_planner_r2c32_with_mode
phast_wisdom_export phast_wisdom_import phast_wisdom_forget phast_wisdom_builtin phast_wisdom_count phast_debug_throw
phast_planner_dit64_describe_call phast_planner_dit32_describe_call phast_planner_r2c64_describe_call phast_planner_r2c32_describe_call
""".split()


class PhastOptions(C.Structure):
    """`phast_options` (options.rs:10-24)."""

    _fields_ = [("multithreaded_bit_reversal", C.c_int), ("smallest_parallel_chunk_size", C.c_size_t)]


class PhastTuneReport(C.Structure):
    """`phast_tune_report` (include/phastft_hip.h: PlannerMode::Tune)."""

    _fields_ = [("adopted", C.c_int), ("candidates", C.c_uint), ("us_heuristic", C.c_float), ("us_best", C.c_float),
                ("seconds", C.c_double), ("plan", C.c_char * 96)]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m phastft_amd.build` (hipcc, gfx950). "
            "phastft_amd has no CPU fallback.")
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's).  Whichever
    # copy is loaded first serves both, and mixing the two leaves the second user without a device.  The package
    # works on torch device tensors and torch streams, so torch's runtime must be the one: import it first.
    try:
        import torch  # noqa: F401
    except ImportError:  # a torch-less process (e.g. a C or Rust host) simply uses /opt/rocm's runtime
        pass
    l = C.CDLL(LIB_PATH)
    l.phast_strerror.restype = C.c_char_p
    l.phast_strerror.argtypes = [C.c_int]
    l.phast_last_hip_error.restype = C.c_char_p
    l.phast_planner_dit64_device_bytes.restype = C.c_size_t
    l.phast_planner_dit32_device_bytes.restype = C.c_size_t
    for name in SYMBOLS:
        getattr(l, name)  # AttributeError here = header/library mismatch
    l.phast_planner_dit64_release_graph_workspaces.restype = C.c_size_t
    l.phast_planner_dit32_release_graph_workspaces.restype = C.c_size_t
    l.phast_wisdom_forget.restype = None
    l.phast_wisdom_builtin.restype = C.c_int
    l.phast_wisdom_builtin.argtypes = [C.c_int]
    l.phast_wisdom_count.restype = C.c_size_t
    l.phast_wisdom_count.argtypes = [C.c_int]
    l.phast_wisdom_import.argtypes = [C.c_char_p]
    l.phast_options_default.restype = None
    l.phast_debug_set_wg_per_cu.restype = None
    l.phast_debug_set_trace.restype = None
    l.phast_debug_set_guard_bytes.restype = None
    l.phast_debug_set_guard_bytes.argtypes = [C.c_size_t]
    for sfx in ("64", "32"):
        getattr(l, f"phast_planner_dit{sfx}_free").restype = None
        getattr(l, f"phast_planner_r2c{sfx}_free").restype = None
        getattr(l, f"phast_twiddle_grid{sfx}_free").restype = None
    _lib = l
    return l
