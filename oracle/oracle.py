"""ctypes front-end of the CPU oracle (oracle/libphastft_oracle.so).

TEST INFRASTRUCTURE ONLY: import this from tests/, ``__graft_entry__.smoke()`` and bench.py's
``cpu_baseline`` leg -- never from the product package ``phastft_amd``.  See phastft_oracle.h.

Function names mirror the reference's Rust API (lib.rs:143-226, algorithms/r2c.rs:521-895,
algorithms/bravo.rs:303-324); panics become :class:`OraclePanic` carrying the reference's message.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libphastft_oracle.so")
_FAST_SO = os.path.join(_HERE, "libphastft_oracle_fast.so")

FORWARD = 1
REVERSE = -1


class OraclePanic(AssertionError):
    """A reference ``assert!``/``assert_eq!`` would have fired (the Rust code panics)."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


def build(force: bool = False) -> str:
    """Compile the oracle with gcc if the .so is missing or older than its sources."""
    srcs = [os.path.join(_HERE, f) for f in ("phastft_oracle.c", "dit_impl.inc", "phastft_oracle.h", "Makefile")]
    fast = os.path.join(_HERE, "libphastft_oracle_fast.so")
    stale = force or not os.path.exists(_SO) or not os.path.exists(fast) or any(
        os.path.getmtime(s) > min(os.path.getmtime(_SO), os.path.getmtime(fast)) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.pho_strerror.restype = C.c_char_p
        _declare_timing(_lib)
        for name in ("pho_planner_dit64_stage", "pho_planner_dit32_stage"):
            getattr(_lib, name).restype = C.c_size_t
    return _lib


def _check(rc: int) -> None:
    if rc != 0:
        raise OraclePanic(rc, lib().pho_strerror(rc).decode())


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _req(a: np.ndarray, dtype) -> np.ndarray:
    if not (isinstance(a, np.ndarray) and a.dtype == dtype and a.flags.c_contiguous and a.ndim == 1):
        raise TypeError(f"need a contiguous 1-D {np.dtype(dtype).name} ndarray")
    return a


def _sz(n: int) -> C.c_size_t:
    return C.c_size_t(n)


# ---- options.rs:38-43 ----
class Options(C.Structure):
    _fields_ = [("multithreaded_bit_reversal", C.c_int), ("smallest_parallel_chunk_size", C.c_size_t)]


def guess_options(input_size: int) -> Options:
    o = Options()
    _check(lib().pho_options_guess(_sz(input_size), C.byref(o)))
    return o


# ---- planners (planner.rs) ----
class _Planner:
    _new = _free = None

    def __init__(self, n: int):
        self._h = C.c_void_p()
        _check(getattr(lib(), self._new)(_sz(n), C.byref(self._h)))
        self.n = n

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value:
            getattr(lib(), self._free)(self._h)
            self._h = C.c_void_p()


class PlannerDit64(_Planner):
    _new, _free, _dtype = "pho_planner_dit64_new", "pho_planner_dit64_free", np.float64

    def stage_twiddles(self, stage: int):
        re, im = C.c_void_p(), C.c_void_p()
        fn = lib().pho_planner_dit64_stage if self._dtype == np.float64 else lib().pho_planner_dit32_stage
        dist = fn(self._h, _sz(stage), C.byref(re), C.byref(im))
        if dist == 0:
            raise IndexError(stage)
        ct = C.c_double if self._dtype == np.float64 else C.c_float
        mk = lambda p: np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(dist,)).copy()
        return mk(re), mk(im)


class PlannerDit32(PlannerDit64):
    _new, _free, _dtype = "pho_planner_dit32_new", "pho_planner_dit32_free", np.float32


class PlannerR2c64(_Planner):
    _new, _free, _dtype = "pho_planner_r2c64_new", "pho_planner_r2c64_free", np.float64

    def twiddles(self):
        re, im = C.c_void_p(), C.c_void_p()
        fn = lib().pho_planner_r2c64_twiddles if self._dtype == np.float64 else lib().pho_planner_r2c32_twiddles
        fn(self._h, C.byref(re), C.byref(im))
        ct = C.c_double if self._dtype == np.float64 else C.c_float
        mk = lambda p: np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(self.n // 2,)).copy()
        return mk(re), mk(im)


class PlannerR2c32(PlannerR2c64):
    _new, _free, _dtype = "pho_planner_r2c32_new", "pho_planner_r2c32_free", np.float32


# ---- C2C (lib.rs:143-226) ----
_flib = None


def fast_lib() -> C.CDLL:
    """The prebuilt -O3 build of the same source (Makefile: `fast`; bit-identical to the checker build,
    tests/test_oracle_pin.py) -- for the parity tests at N >= 2^23, where the -O2 checker takes several seconds per
    transform.  Planner handles are plain heap structs of the same source: one made by lib() serves here too."""
    global _flib
    if _flib is None:
        build()
        _flib = C.CDLL(_FAST_SO)
        _flib.pho_strerror.restype = C.c_char_p
    return _flib


def _fft(name, dtype, reals, imags, direction, planner=None, fast=False):
    _req(reals, dtype), _req(imags, dtype)
    l = fast_lib() if fast else lib()
    if planner is None:
        _check(getattr(l, name)(_p(reals), _sz(reals.size), _p(imags), _sz(imags.size), C.c_int(direction)))
    else:
        _check(getattr(l, name + "_with_planner")(_p(reals), _sz(reals.size), _p(imags), _sz(imags.size),
                                                  C.c_int(direction), planner._h))


def fft_64_dit(reals, imags, direction=FORWARD):
    _fft("pho_fft_64_dit", np.float64, reals, imags, direction)


def fft_32_dit(reals, imags, direction=FORWARD):
    _fft("pho_fft_32_dit", np.float32, reals, imags, direction)


def fft_64_dit_with_planner_parallel(reals, imags, direction, planner: PlannerDit64):
    """feature ``parallel`` emulated (rayon::join -> OpenMP tasks); bit-identical to the serial call."""
    _req(reals, np.float64), _req(imags, np.float64)
    _check(lib().pho_fft_64_dit_with_planner_parallel(_p(reals), _sz(reals.size), _p(imags), _sz(imags.size),
                                                      C.c_int(direction), planner._h))


def fft_32_dit_with_planner_parallel(reals, imags, direction, planner: PlannerDit32):
    _req(reals, np.float32), _req(imags, np.float32)
    _check(lib().pho_fft_32_dit_with_planner_parallel(_p(reals), _sz(reals.size), _p(imags), _sz(imags.size),
                                                      C.c_int(direction), planner._h))


def fft_64_dit_with_planner(reals, imags, direction, planner: PlannerDit64, fast: bool = False):
    _fft("pho_fft_64_dit", np.float64, reals, imags, direction, planner, fast)


def fft_32_dit_with_planner(reals, imags, direction, planner: PlannerDit32, fast: bool = False):
    _fft("pho_fft_32_dit", np.float32, reals, imags, direction, planner, fast)


# ---- bit reversal (bravo.rs) ----
def bit_rev_bravo_f64(data, log_n, regime=""):
    _req(data, np.float64)
    assert data.size == 1 << log_n, "Data length must be 2^n"
    getattr(lib(), "pho_bit_rev" + regime + "_f64")(_p(data), C.c_uint(log_n))


def bit_rev_bravo_f32(data, log_n, regime=""):
    _req(data, np.float32)
    assert data.size == 1 << log_n, "Data length must be 2^n"
    getattr(lib(), "pho_bit_rev" + regime + "_f32")(_p(data), C.c_uint(log_n))


# ---- Complex<T> <-> planes (complex_nums.rs:11-56) ----
def deinterleave(data):
    """complex_nums.rs:11: (a, b) = (data[0::2], data[1::2]) over the complete pairs"""
    fs = "f64" if data.dtype == np.float64 else "f32"
    _req(data, data.dtype.type)
    a, b = np.empty(data.size // 2, data.dtype), np.empty(data.size // 2, data.dtype)
    getattr(lib(), "pho_deinterleave_" + fs)(_p(data), C.c_size_t(data.size), _p(a), _p(b))
    return a, b


def combine_re_im(reals, imags):
    """complex_nums.rs:47: the 2 n scalars of the Complex<T> array"""
    assert reals.size == imags.size  # complex_nums.rs:48
    fs = "f64" if reals.dtype == np.float64 else "f32"
    out = np.empty(2 * reals.size, reals.dtype)
    getattr(lib(), "pho_combine_re_im_" + fs)(_p(reals), _p(imags), C.c_size_t(reals.size), _p(out))
    return out


# ---- kernels for the pin tests ----
def codelet_16_f64(re, im):
    lib().pho_codelet_16_f64(_p(_req(re, np.float64)), _p(_req(im, np.float64)), _sz(re.size))


def codelet_32_f32(re, im):
    lib().pho_codelet_32_f32(_p(_req(re, np.float32)), _p(_req(im, np.float32)), _sz(re.size))


def stage_const(re, im, stage):
    sfx = "f64" if re.dtype == np.float64 else "f32"
    getattr(lib(), "pho_stage_const_" + sfx)(_p(re), _p(im), _sz(re.size), C.c_uint(stage))


# ---- R2C / C2R (r2c.rs) ----
def _r2c(sfx, dtype, input_re, output_re, output_im, planner=None):
    _req(input_re, dtype), _req(output_re, dtype), _req(output_im, dtype)
    args = [_p(input_re), _sz(input_re.size), _p(output_re), _sz(output_re.size), _p(output_im), _sz(output_im.size)]
    if planner is None:
        _check(getattr(lib(), "pho_r2c_fft_" + sfx)(*args))
    else:
        _check(getattr(lib(), "pho_r2c_fft_" + sfx + "_with_planner")(*args, planner._h))


def r2c_fft_f64(input_re, output_re, output_im):
    _r2c("f64", np.float64, input_re, output_re, output_im)


def r2c_fft_f32(input_re, output_re, output_im):
    _r2c("f32", np.float32, input_re, output_re, output_im)


def r2c_fft_f64_with_planner(input_re, output_re, output_im, planner):
    _r2c("f64", np.float64, input_re, output_re, output_im, planner)


def r2c_fft_f32_with_planner(input_re, output_re, output_im, planner):
    _r2c("f32", np.float32, input_re, output_re, output_im, planner)


def _c2r(sfx, dtype, input_re, input_im, output, planner=None, scratch_re=None, scratch_im=None):
    _req(input_re, dtype), _req(input_im, dtype), _req(output, dtype)
    args = [_p(input_re), _sz(input_re.size), _p(input_im), _sz(input_im.size), _p(output), _sz(output.size)]
    if planner is None:
        _check(getattr(lib(), "pho_c2r_fft_" + sfx)(*args))
        return
    if scratch_re is None:  # r2c.rs:710-725: the allocating form
        half = planner.n // 2
        scratch_re, scratch_im = np.zeros(half, dtype), np.zeros(half, dtype)
    _req(scratch_re, dtype), _req(scratch_im, dtype)
    _check(getattr(lib(), "pho_c2r_fft_" + sfx + "_with_planner_and_scratch")(
        *args, planner._h, _p(scratch_re), _sz(scratch_re.size), _p(scratch_im), _sz(scratch_im.size)))


def c2r_fft_f64(input_re, input_im, output):
    _c2r("f64", np.float64, input_re, input_im, output)


def c2r_fft_f32(input_re, input_im, output):
    _c2r("f32", np.float32, input_re, input_im, output)


def c2r_fft_f64_with_planner(input_re, input_im, output, planner):
    _c2r("f64", np.float64, input_re, input_im, output, planner)


def c2r_fft_f32_with_planner(input_re, input_im, output, planner):
    _c2r("f32", np.float32, input_re, input_im, output, planner)


def c2r_fft_f64_with_planner_and_scratch(input_re, input_im, output, planner, scratch_re, scratch_im):
    _c2r("f64", np.float64, input_re, input_im, output, planner, scratch_re, scratch_im)


def c2r_fft_f32_with_planner_and_scratch(input_re, input_im, output, planner, scratch_re, scratch_im):
    _c2r("f32", np.float32, input_re, input_im, output, planner, scratch_re, scratch_im)


# ---- synthetic inputs + timing ----
def fill(n: int, dtype, seed: int = 0xCAFE, transform_id: int = 0):
    re, im = np.empty(n, dtype), np.empty(n, dtype)
    sfx = "f64" if np.dtype(dtype) == np.float64 else "f32"
    getattr(lib(), "pho_fill_" + sfx)(_p(re), _p(im), _sz(n), C.c_ulonglong(seed), C.c_ulonglong(transform_id))
    return re, im


# The timing legs run in the TIMING build of the same source (-O3, vectorised butterfly loops; Makefile: `fast`, or
# `native` = -march=native compiled on the box that runs the benchmark).  tests/test_oracle_pin.py checks that it
# returns bit-identical results to the checker build above.
_NATIVE_SO = os.path.join(_HERE, "_native", "libphastft_oracle_native.so")
_tlib = None
_tkind = ""


def _declare_timing(l: C.CDLL) -> C.CDLL:
    for name, args in (("pho_time_fft_64_dit", [C.c_size_t, C.c_int, C.c_ulonglong]),
                       ("pho_time_fft_32_dit", [C.c_size_t, C.c_int, C.c_ulonglong]),
                       ("pho_time_fft_64_dit_parallel", [C.c_size_t, C.c_int, C.c_ulonglong, C.c_int]),
                       ("pho_time_fft_64_roundtrip", [C.c_size_t, C.c_int, C.c_ulonglong]),
                       ("pho_time_r2c_fft_f32", [C.c_size_t, C.c_int, C.c_ulonglong]),
                       ("pho_time_c2r_fft_f32", [C.c_size_t, C.c_int, C.c_ulonglong])):
        getattr(l, name).restype = C.c_double
        getattr(l, name).argtypes = args
    return l


def timing_lib(native: bool = True) -> C.CDLL:
    """The -O3 build for the cpu_baseline legs: `make native` on this host when gcc is here (the build is keyed to
    this CPU and never travels: oracle/_native/ is git- and gpurun-ignored), else the prebuilt AVX2+FMA one."""
    global _tlib, _tkind
    if _tlib is None:
        build()
        if native:
            try:
                subprocess.run(["make", "-C", _HERE, "-B", "native"], check=True, capture_output=True, timeout=120)
                _tlib, _tkind = _declare_timing(C.CDLL(_NATIVE_SO)), "-O3 -march=native"
            except (OSError, subprocess.SubprocessError):
                _tlib = None
        if _tlib is None:
            _tlib, _tkind = _declare_timing(C.CDLL(_FAST_SO)), "-O3 -mavx2 -mfma"
    return _tlib


def timing_build() -> str:
    timing_lib()
    return _tkind


def time_fft_64_dit(n: int, iters: int, seed: int = 0xCAFE) -> float:
    return timing_lib().pho_time_fft_64_dit(n, iters, seed)


def time_fft_32_dit(n: int, iters: int, seed: int = 0xCAFE) -> float:
    return timing_lib().pho_time_fft_32_dit(n, iters, seed)


def time_fft_64_dit_parallel(n: int, iters: int, seed: int = 0xCAFE, threads: int = 0) -> float:
    return timing_lib().pho_time_fft_64_dit_parallel(n, iters, seed, threads)


def time_fft_64_roundtrip(n: int, iters: int, seed: int = 0xCAFE) -> float:
    return timing_lib().pho_time_fft_64_roundtrip(n, iters, seed)


def parallel_threads() -> int:
    return lib().pho_parallel_threads()


def time_r2c_fft_f32(n: int, iters: int, seed: int = 0xCAFE) -> float:
    return timing_lib().pho_time_r2c_fft_f32(n, iters, seed)


def time_c2r_fft_f32(n: int, iters: int, seed: int = 0xCAFE) -> float:
    return timing_lib().pho_time_c2r_fft_f32(n, iters, seed)
