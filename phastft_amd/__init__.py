"""phastft_amd -- MI355X (gfx950) drop-in for PhastFT's planar power-of-two FFT path.

Host-side mirror of the reference's public Rust API (QuState/PhastFT 0.3.0; citations are relative to
the reference tree) over the C ABI of ``include/phastft_hip.h``:

    ==============================================  ==========================================
    reference (Rust)                                here
    ==============================================  ==========================================
    Direction, PlannerMode            planner.rs:10  Direction, PlannerMode
    Options, Options::guess_options   options.rs:10  Options, Options.guess_options
    PlannerDit64/32::{new,with_mode}  planner.rs:55  PlannerDit64/32(n), .with_mode(n, mode)
    PlannerR2c64/32::new              planner.rs:194 PlannerR2c64/32(n)
    fft_64_dit / fft_32_dit           lib.rs:180,223 fft_64_dit / fft_32_dit
    fft_*_dit_with_planner[_and_opts] lib.rs:143,186 same names
    r2c_fft_f32/f64[_with_planner]    r2c.rs:521-662 same names
    c2r_fft_*[_with_planner[_and_scratch]] r2c.rs:695 same names
    bit_rev_bravo_f32/f64             bravo.rs:303   bit_rev_bravo_f32/f64(data, n)
    deinterleave[_complex64/32]       complex_nums.rs:11   deinterleave(data) -> (a, b)
    combine_re_im                     complex_nums.rs:47   combine_re_im(reals, imags) -> Complex<T> array
    ==============================================  ==========================================

Slices are 1-D contiguous arrays: ``numpy.ndarray`` (host slices -- staged through device memory, the
Rust drop-in semantics) or ``torch.Tensor`` on ``cuda`` (device-resident, asynchronous on torch's current
stream; this is the measured path).  All transforms are in place on planar ``reals`` / ``imags``; forward is
unnormalised and ``Direction.Reverse`` scales by 1/N, as in the reference (README.md:167-172).

The reference signals misuse by panicking; here every reference ``assert!`` raises :class:`PhastPanic`
carrying the reference's message.  There is NO CPU fallback: without the HIP library or a GPU the calls
raise.
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass

import numpy as np

from . import _lib

__all__ = [
    "Direction", "PlannerMode", "TuneKind", "wisdom_export", "wisdom_import", "wisdom_forget", "wisdom_builtin", "wisdom_count", "Options", "PhastPanic", "PhastHipError",
    "PlannerDit64", "PlannerDit32", "PlannerR2c64", "PlannerR2c32",
    "fft_64_dit", "fft_32_dit", "fft_64_dit_with_planner", "fft_32_dit_with_planner",
    "fft_64_dit_with_planner_and_opts", "fft_32_dit_with_planner_and_opts",
    "r2c_fft_f64", "r2c_fft_f32", "r2c_fft_f64_with_planner", "r2c_fft_f32_with_planner",
    "c2r_fft_f64", "c2r_fft_f32", "c2r_fft_f64_with_planner", "c2r_fft_f32_with_planner",
    "c2r_fft_f64_with_planner_and_scratch", "c2r_fft_f32_with_planner_and_scratch",
    "fft_dit_strided", "fft_64_interleaved", "fft_32_interleaved", "fft_64_interleaved_with_planner", "fft_32_interleaved_with_planner",
    "fft_64_interleaved_with_planner_and_opts", "fft_32_interleaved_with_planner_and_opts",
    "bit_rev_bravo_f64", "bit_rev_bravo_f32", "deinterleave", "deinterleave_complex64", "deinterleave_complex32", "combine_re_im", "fft_dit_batched", "r2c_fft_batched", "c2r_fft_batched", "fill_uniform", "digest", "device_info",
    "TwiddleGrid64", "TwiddleGrid32",
]


class Direction(enum.IntEnum):
    """planner.rs:10-16"""

    Forward = 1
    Reverse = -1


class PlannerMode(enum.IntEnum):
    """planner.rs:24-32.  ``Tune``: the planner times the plans that exist for its length on the device at plan time and
    keeps the fastest (one transform per call; :meth:`PlannerDit64.tune` for other batch sizes and call kinds)."""

    Heuristic = 0
    Tune = 1


class TuneKind(enum.IntEnum):
    """PHAST_TUNE_* (include/phastft_hip.h): which call a tuning run measures."""

    C2C = 0
    C2CInterleaved = 1
    R2C = 2
    C2R = 3


def _tune(fn, handle, batch: int, kind: "TuneKind") -> dict:
    rep = _lib.PhastTuneReport()
    _check(fn(handle, C.c_size_t(batch), C.c_int(int(kind)), C.byref(rep)))
    return {"adopted": bool(rep.adopted), "candidates": int(rep.candidates), "us_heuristic": float(rep.us_heuristic),
            "us_best": float(rep.us_best), "seconds": float(rep.seconds), "plan": rep.plan.decode()}


def wisdom_count(layer: int = -1) -> int:
    """Entries of a wisdom layer: 0 built-in, 1 PHAST_WISDOM file, 2 imported, 3 measured by this process; -1 all."""
    return int(_lib.lib().phast_wisdom_count(C.c_int(layer)))


def wisdom_export() -> str:
    """What this process measured, imported or read from PHAST_WISDOM, as text (csrc/wisdom.hpp) -- without the built-in layer."""
    need = C.c_size_t(0)
    _check(_lib.lib().phast_wisdom_export(None, C.c_size_t(0), C.byref(need)))
    buf = C.create_string_buffer(need.value)
    _check(_lib.lib().phast_wisdom_export(buf, C.c_size_t(need.value), None))
    return buf.value.decode()


def wisdom_import(text: str) -> None:
    """Planners created afterwards start with the plans the text names."""
    _check(_lib.lib().phast_wisdom_import(text.encode()))


def wisdom_forget() -> None:
    _lib.lib().phast_wisdom_forget()


def wisdom_builtin(enable: bool) -> bool:
    """The wisdom compiled into the library (csrc/builtin_wisdom.inc) off / on for planners made afterwards.  Returns what it
    was before (PHAST_BUILTIN_WISDOM=0 starts it off): `was = wisdom_builtin(False) ... wisdom_builtin(was)` puts it back."""
    return bool(_lib.lib().phast_wisdom_builtin(1 if enable else 0))


ERR_INVALID_ARG = 16  # PHAST_ERR_INVALID_ARG (include/phastft_hip.h): e.g. a shape the strided kernels do not cover


class PhastPanic(AssertionError):
    """A reference ``assert!`` / ``assert_eq!`` would have fired; ``str(e)`` is the reference's message."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


class PhastHipError(RuntimeError):
    """The HIP runtime failed or no GPU is visible (codes 14/15 of phastft_hip.h)."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


def _check(rc: int) -> None:
    if rc == 0:
        return
    l = _lib.lib()
    msg = l.phast_strerror(rc).decode()
    if rc in (13, 14, 15):
        raise PhastHipError(rc, f"{msg}: {l.phast_last_hip_error().decode()}")
    raise PhastPanic(rc, msg)


@dataclass
class Options:
    """options.rs:8-43.  CPU threading knobs: carried for source compatibility, ignored on the GPU."""

    multithreaded_bit_reversal: bool = False
    smallest_parallel_chunk_size: int = 16384

    @staticmethod
    def guess_options(input_size: int) -> "Options":
        o = _lib.PhastOptions()
        _check(_lib.lib().phast_options_guess(C.c_size_t(input_size), C.byref(o)))
        return Options(bool(o.multithreaded_bit_reversal), int(o.smallest_parallel_chunk_size))

    def _c(self) -> _lib.PhastOptions:
        return _lib.PhastOptions(int(self.multithreaded_bit_reversal), self.smallest_parallel_chunk_size)


# ---------------------------------------------------------------------------------------------
# slices
# ---------------------------------------------------------------------------------------------
def _is_torch(x) -> bool:
    return type(x).__module__.split(".")[0] == "torch"


class _Slice:
    """pointer + length + where it lives, for a numpy array or a torch tensor"""

    __slots__ = ("ptr", "len", "dev", "keep")

    def __init__(self, x, dtype, what: str):
        if _is_torch(x):
            import torch

            want = torch.float64 if dtype == np.float64 else torch.float32
            if x.dtype != want or x.dim() != 1 or not x.is_contiguous():
                raise TypeError(f"{what}: need a contiguous 1-D {want} tensor")
            self.dev = x.device.type == "cuda"
            if not self.dev:
                raise TypeError(f"{what}: torch tensors must live on the GPU (use numpy arrays for host slices)")
            self.ptr = C.c_void_p(x.data_ptr())
            self.len = x.numel()
        elif isinstance(x, np.ndarray):
            if x.dtype != dtype or x.ndim != 1 or not x.flags.c_contiguous:
                raise TypeError(f"{what}: need a contiguous 1-D {np.dtype(dtype).name} ndarray")
            self.dev = False
            self.ptr = x.ctypes.data_as(C.c_void_p)
            self.len = x.size
        else:
            raise TypeError(f"{what}: need a numpy.ndarray (host) or a torch cuda tensor (device)")
        self.keep = x


def _stream() -> C.c_void_p:
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _same_place(*slices: _Slice) -> bool:
    dev = slices[0].dev
    if any(s.dev != dev for s in slices):
        raise TypeError("all slices of one call must be host arrays or all device tensors")
    return dev


# ---------------------------------------------------------------------------------------------
# planners
# ---------------------------------------------------------------------------------------------
class _PlannerDit:
    _sfx = "64"
    _dtype = np.float64

    def __init__(self, num_points: int, mode: PlannerMode = PlannerMode.Heuristic):
        self._h = C.c_void_p()
        l = _lib.lib()
        _check(getattr(l, f"phast_planner_dit{self._sfx}_with_mode")(C.c_size_t(num_points), C.c_int(int(mode)),
                                                                     C.byref(self._h)))
        self.num_points = num_points

    @classmethod
    def new(cls, num_points: int):
        """planner.rs:55"""
        return cls(num_points)

    @classmethod
    def with_mode(cls, num_points: int, mode: PlannerMode):
        """planner.rs:65"""
        return cls(num_points, mode)

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value and _lib is not None and _lib._lib is not None:
                getattr(_lib._lib, f"phast_planner_dit{self._sfx}_free")(self._h)
                self._h.value = None
        except Exception:  # interpreter shutdown: modules may already be gone
            pass

    # ---- MI355X-side extras (no reference counterpart) ----
    def describe(self) -> str:
        buf = C.create_string_buffer(16384)   # (a planner may carry a dozen wisdom plans besides its static ones)
        _check(getattr(_lib.lib(), f"phast_planner_dit{self._sfx}_describe")(self._h, buf, C.c_size_t(16384)))
        return buf.value.decode()

    def device_bytes(self) -> int:
        return int(getattr(_lib.lib(), f"phast_planner_dit{self._sfx}_device_bytes")(self._h))

    def describe_call(self, batch: int = 1, kind: "TuneKind" = 0) -> str:
        """The plan a call with ``batch`` transforms runs: ``"<which> [rows x cols ...]..."`` (the library's own answer)."""
        buf = C.create_string_buffer(512)
        _check(getattr(_lib.lib(), f"phast_planner_dit{self._sfx}_describe_call")(self._h, C.c_size_t(batch), C.c_int(int(kind)), buf, C.c_size_t(512)))
        return buf.value.decode()

    def check_guards(self) -> int:
        """debug: bytes of the scratch's guard bands overwritten since allocation (see :func:`debug_set_guard_bytes`)"""
        bad = C.c_size_t(0)
        _check(getattr(_lib.lib(), f"phast_planner_dit{self._sfx}_debug_check_guards")(self._h, C.byref(bad)))
        return int(bad.value)

    def reserve_batch(self, max_batch: int) -> None:
        _check(getattr(_lib.lib(), f"phast_planner_dit{self._sfx}_reserve_batch")(self._h, C.c_size_t(max_batch)))

    def release_graph_workspaces(self) -> int:
        """Hand back the workspaces captured graphs worked in (every graph captured on this planner must be gone)."""
        return int(getattr(_lib.lib(), f"phast_planner_dit{self._sfx}_release_graph_workspaces")(self._h))

    def tune(self, batch: int = 1, kind: TuneKind = TuneKind.C2C) -> dict:
        """PlannerMode::Tune for ``batch`` transforms per call (covers batches in (2^(b-1), 2^b]) and the call kind
        ``TuneKind.C2C`` / ``C2CInterleaved``: measures on the device, installs the winner, returns the report."""
        return _tune(getattr(_lib.lib(), f"phast_planner_dit{self._sfx}_tune"), self._h, batch, kind)

    def set_plan(self, log_rows=(), tile_log=12, points_log=4) -> None:
        """Force the pass factorisation (tuning hook); ``()`` restores the heuristic.  ``tile_log`` is
        log2(points per tile): one int for all passes or one per pass; ``points_log`` = log2(points per thread)."""
        n = len(log_rows)
        tls = [tile_log] * n if isinstance(tile_log, int) else list(tile_log)
        arr = (C.c_uint * max(1, n))(*log_rows)
        tarr = (C.c_uint * max(1, n))(*tls)
        _check(getattr(_lib.lib(), f"phast_planner_dit{self._sfx}_set_plan")(self._h, arr, tarr, C.c_size_t(n), C.c_uint(points_log)))


    def time_passes(self, reals, imags, n: int, reps: int = 10):
        """Average HIP-event duration (ms) of every pass kernel over ``reps`` forward transforms of the
        device tensors (``len/n`` transforms, transformed in place).  Measurement hook for bench.py."""
        re, im = _Slice(reals, self._dtype, "reals"), _Slice(imags, self._dtype, "imags")
        ms = (C.c_float * 3)()
        npass = C.c_int()
        _check(getattr(_lib.lib(), f"phast_planner_dit{self._sfx}_time_passes")(
            self._h, re.ptr, im.ptr, C.c_size_t(re.len // n), C.c_size_t(n), C.c_int(reps), ms, C.byref(npass),
            _stream()))
        return [float(ms[i]) for i in range(npass.value)]


class PlannerDit64(_PlannerDit):
    """planner.rs:34-114 (f64)"""


class PlannerDit32(_PlannerDit):
    """planner.rs:34-114 (f32)"""

    _sfx = "32"
    _dtype = np.float32


class _PlannerR2c:
    _sfx = "64"
    _dtype = np.float64

    def __init__(self, n: int, mode: PlannerMode = PlannerMode.Heuristic):
        self._h = C.c_void_p()
        _check(getattr(_lib.lib(), f"phast_planner_r2c{self._sfx}_with_mode")(C.c_size_t(n), C.c_int(int(mode)), C.byref(self._h)))
        self.n = n

    @classmethod
    def new(cls, n: int):
        """planner.rs:194"""
        return cls(n)

    @classmethod
    def with_mode(cls, n: int, mode: PlannerMode):
        """(no reference counterpart: the PlannerDit*::with_mode switch for the real transforms)"""
        return cls(n, mode)

    def describe_call(self, batch: int = 1, kind: "TuneKind" = 2) -> str:
        """The plan of the inner transform an ``r2c_fft`` / ``c2r_fft`` call with ``batch`` transforms runs."""
        buf = C.create_string_buffer(512)
        _check(getattr(_lib.lib(), f"phast_planner_r2c{self._sfx}_describe_call")(self._h, C.c_size_t(batch), C.c_int(int(kind)), buf, C.c_size_t(512)))
        return buf.value.decode()

    def tune(self, batch: int = 1, kind: TuneKind = TuneKind.R2C) -> dict:
        """PlannerMode::Tune for ``batch`` real transforms per call, ``TuneKind.R2C`` or ``TuneKind.C2R``."""
        return _tune(getattr(_lib.lib(), f"phast_planner_r2c{self._sfx}_tune"), self._h, batch, kind)

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value and _lib is not None and _lib._lib is not None:
                getattr(_lib._lib, f"phast_planner_r2c{self._sfx}_free")(self._h)
                self._h.value = None
        except Exception:  # interpreter shutdown: modules may already be gone
            pass


    def describe(self) -> str:
        """Plan of the inner N/2-point complex transform."""
        buf = C.create_string_buffer(16384)
        _check(getattr(_lib.lib(), f"phast_planner_r2c{self._sfx}_describe")(self._h, buf, C.c_size_t(16384)))
        return buf.value.decode()

    def set_plan(self, log_rows=(), tile_log=12, points_log=4) -> None:
        """Force the pass factorisation of the inner N/2-point transform (tuning hook, as ``PlannerDit*.set_plan``);
        ``()`` restores the library's own plans."""
        n = len(log_rows)
        tls = [tile_log] * n if isinstance(tile_log, int) else list(tile_log)
        arr = (C.c_uint * max(1, n))(*log_rows)
        tarr = (C.c_uint * max(1, n))(*tls)
        _check(getattr(_lib.lib(), f"phast_planner_r2c{self._sfx}_set_inner_plan")(self._h, arr, tarr, C.c_size_t(n), C.c_uint(points_log)))

    def time_passes(self, input_re, output_re, output_im, reps: int = 10):
        """Average HIP-event duration (ms) of every kernel of one R2C transform of the device tensors: the passes of the
        inner N/2-point transform, then the untangle sweep.  Measurement hook for bench.py."""
        i, ore, oim = (_Slice(x, self._dtype, w) for x, w in ((input_re, "input_re"), (output_re, "output_re"),
                                                              (output_im, "output_im")))
        ms = (C.c_float * 4)()
        npass = C.c_int()
        _check(getattr(_lib.lib(), f"phast_planner_r2c{self._sfx}_time_passes")(
            self._h, i.ptr, ore.ptr, oim.ptr, C.c_size_t(1), C.c_size_t(self.n), C.c_size_t(self.n // 2 + 1),
            C.c_int(reps), ms, C.byref(npass), _stream()))
        return [float(ms[k]) for k in range(npass.value)]


    def time_c2r_passes(self, input_re, input_im, output, reps: int = 10):
        """The same for one C2R transform: the passes of the inner transform (the first one forms z on load where its
        fused form exists), then the preprocess sweep where it does not."""
        ire, iim, out = (_Slice(x, self._dtype, w) for x, w in ((input_re, "input_re"), (input_im, "input_im"), (output, "output")))
        ms = (C.c_float * 4)()
        npass = C.c_int()
        _check(getattr(_lib.lib(), f"phast_planner_r2c{self._sfx}_time_c2r_passes")(
            self._h, ire.ptr, iim.ptr, out.ptr, C.c_size_t(1), C.c_size_t(self.n // 2 + 1), C.c_size_t(self.n),
            C.c_int(reps), ms, C.byref(npass), _stream()))
        return [float(ms[k]) for k in range(npass.value)]


class PlannerR2c64(_PlannerR2c):
    """planner.rs:164-212 (f64)"""


class PlannerR2c32(_PlannerR2c):
    """planner.rs:164-212 (f32)"""

    _sfx = "32"
    _dtype = np.float32


# ---------------------------------------------------------------------------------------------
# C2C  (lib.rs:143-226, algorithms/dit.rs:263,338)
# ---------------------------------------------------------------------------------------------
def _fft(sfx, dtype, reals, imags, direction, planner=None, opts=None, need_opts=False):
    re, im = _Slice(reals, dtype, "reals"), _Slice(imags, dtype, "imags")
    l = _lib.lib()
    direction = C.c_int(int(direction))
    if _same_place(re, im):
        # device-resident: the Rust asserts are re-checked here, then the batched _dev entry point is used
        own = planner is None
        if own:
            planner = (PlannerDit64 if sfx == "64" else PlannerDit32)(re.len)  # lib.rs:181: planner from reals.len()
        if re.len != im.len:
            _check(2)
        _check(getattr(l, f"phast_fft_{sfx}_dit_dev")(re.ptr, im.ptr, C.c_size_t(re.len), C.c_size_t(1),
                                                      C.c_size_t(re.len), direction, planner._h, _stream()))
        if own:
            import torch

            torch.cuda.current_stream().synchronize()  # the temporary planner's scratch dies with it
        return
    args = [re.ptr, C.c_size_t(re.len), im.ptr, C.c_size_t(im.len), direction]
    if planner is None:
        _check(getattr(l, f"phast_fft_{sfx}_dit")(*args))
    elif need_opts:
        _check(getattr(l, f"phast_fft_{sfx}_dit_with_planner_and_opts")(*args, planner._h, C.byref(opts._c())))
    else:
        _check(getattr(l, f"phast_fft_{sfx}_dit_with_planner")(*args, planner._h))


def fft_64_dit(reals, imags, direction: Direction) -> None:
    """lib.rs:180"""
    _fft("64", np.float64, reals, imags, direction)


def fft_32_dit(reals, imags, direction: Direction) -> None:
    """lib.rs:223"""
    _fft("32", np.float32, reals, imags, direction)


def fft_64_dit_with_planner(reals, imags, direction: Direction, planner: PlannerDit64) -> None:
    """lib.rs:143"""
    _fft("64", np.float64, reals, imags, direction, planner)


def fft_32_dit_with_planner(reals, imags, direction: Direction, planner: PlannerDit32) -> None:
    """lib.rs:186"""
    _fft("32", np.float32, reals, imags, direction, planner)


def fft_64_dit_with_planner_and_opts(reals, imags, direction: Direction, planner: PlannerDit64, opts: Options) -> None:
    """algorithms/dit.rs:263"""
    _fft("64", np.float64, reals, imags, direction, planner, opts, True)


def fft_32_dit_with_planner_and_opts(reals, imags, direction: Direction, planner: PlannerDit32, opts: Options) -> None:
    """algorithms/dit.rs:338"""
    _fft("32", np.float32, reals, imags, direction, planner, opts, True)


# ---------------------------------------------------------------------------------------------
# interleaved Complex<T> signals  (lib.rs:41-140, feature `complex-nums`)
# ---------------------------------------------------------------------------------------------
def _fft_interleaved(sfx, cdtype, signal, direction, planner=None, opts=None, need_opts=False):
    l = _lib.lib()
    fdtype = np.float64 if sfx == "64" else np.float32
    if _is_torch(signal):
        import torch

        want = torch.complex128 if sfx == "64" else torch.complex64
        if signal.dtype != want or signal.dim() != 1 or not signal.is_contiguous() or signal.device.type != "cuda":
            raise TypeError(f"signal: need a contiguous 1-D {want} cuda tensor")
        n = signal.numel()
        own = planner is None
        if own:
            planner = (PlannerDit64 if sfx == "64" else PlannerDit32)(n)
        _check(getattr(l, f"phast_fft_{sfx}_interleaved_dev")(C.c_void_p(signal.data_ptr()), C.c_size_t(n), C.c_size_t(1),
                                                              C.c_size_t(n), C.c_int(int(direction)), planner._h,
                                                              _stream()))
        if own:
            torch.cuda.current_stream().synchronize()
        return
    if not (isinstance(signal, np.ndarray) and signal.dtype == cdtype and signal.ndim == 1 and signal.flags.c_contiguous):
        raise TypeError(f"signal: need a contiguous 1-D {np.dtype(cdtype).name} ndarray or a cuda tensor")
    flat = signal.view(fdtype)
    args = [flat.ctypes.data_as(C.c_void_p), C.c_size_t(signal.size), C.c_int(int(direction))]
    if planner is None:
        _check(getattr(l, f"phast_fft_{sfx}_interleaved")(*args))
    elif need_opts:
        _check(getattr(l, f"phast_fft_{sfx}_interleaved_with_planner_and_opts")(*args, planner._h, C.byref(opts._c())))
    else:
        _check(getattr(l, f"phast_fft_{sfx}_interleaved_with_planner")(*args, planner._h))


def fft_64_interleaved(signal, direction: Direction) -> None:
    """lib.rs:120 (macro impl_fft_interleaved)"""
    _fft_interleaved("64", np.complex128, signal, direction)


def fft_32_interleaved(signal, direction: Direction) -> None:
    """lib.rs:120"""
    _fft_interleaved("32", np.complex64, signal, direction)


def fft_64_interleaved_with_planner(signal, direction: Direction, planner: PlannerDit64) -> None:
    """lib.rs:87"""
    _fft_interleaved("64", np.complex128, signal, direction, planner)


def fft_32_interleaved_with_planner(signal, direction: Direction, planner: PlannerDit32) -> None:
    """lib.rs:87"""
    _fft_interleaved("32", np.complex64, signal, direction, planner)


def fft_64_interleaved_with_planner_and_opts(signal, direction: Direction, planner: PlannerDit64, opts: Options) -> None:
    """lib.rs:50"""
    _fft_interleaved("64", np.complex128, signal, direction, planner, opts, True)


def fft_32_interleaved_with_planner_and_opts(signal, direction: Direction, planner: PlannerDit32, opts: Options) -> None:
    """lib.rs:50"""
    _fft_interleaved("32", np.complex64, signal, direction, planner, opts, True)


def fft_dit_batched(reals, imags, n: int, direction: Direction, planner, dist: int | None = None) -> None:
    """Device-resident batch: transform b lives at ``[b*dist, b*dist + n)`` of ``reals``/``imags`` (``dist`` defaults
    to ``n``: transforms back to back).  No reference counterpart -- the reference loops over transforms on the CPU."""
    dtype, sfx = planner._dtype, planner._sfx
    re, im = _Slice(reals, dtype, "reals"), _Slice(imags, dtype, "imags")
    if not _same_place(re, im):
        raise TypeError("fft_dit_batched needs device tensors")
    if re.len != im.len:
        _check(2)
    dist = n if dist is None else dist
    if n == 0 or dist < n or re.len < n or (re.len - n) % dist:
        raise ValueError("length must be (batch-1)*dist + n")
    batch = (re.len - n) // dist + 1
    _check(getattr(_lib.lib(), f"phast_fft_{sfx}_dit_dev")(re.ptr, im.ptr, C.c_size_t(n), C.c_size_t(batch),
                                                           C.c_size_t(dist), C.c_int(int(direction)), planner._h,
                                                           _stream()))


class TransformList:
    """`count` independent transforms at arbitrary device addresses, prepared once (the pointer arrays) and enqueued by
    ONE call into the library -- each runs exactly as a single-transform call (``phast_fft_*_dit_many_dev``)."""

    def __init__(self, pairs, n: int, planner):
        dtype = planner._dtype
        self._keep = [(_Slice(r, dtype, "reals"), _Slice(m, dtype, "imags")) for r, m in pairs]
        for r, m in self._keep:
            if not (r.dev and m.dev) or r.len != n or m.len != n:
                raise TypeError("TransformList needs device tensors of n elements each")
        k = len(self._keep)
        self._re = (C.c_void_p * k)(*[r.ptr.value for r, _ in self._keep])
        self._im = (C.c_void_p * k)(*[m.ptr.value for _, m in self._keep])
        self.n, self.count, self.planner = n, k, planner

    def run(self, direction: Direction, first: int = 0, count: int | None = None) -> None:
        k = self.count - first if count is None else count
        if first < 0 or k < 0 or first + k > self.count:
            raise ValueError("range outside the list")
        off = first * C.sizeof(C.c_void_p)
        _check(getattr(_lib.lib(), f"phast_fft_{self.planner._sfx}_dit_many_dev")(
            C.c_void_p(C.addressof(self._re) + off), C.c_void_p(C.addressof(self._im) + off), C.c_size_t(k),
            C.c_size_t(self.n), C.c_int(int(direction)), self.planner._h, _stream()))


def fft_dit_strided(reals, imags, n: int, direction: Direction, planner, batch: int, stride: int,
                    twiddle_n: int = 0, twiddle_col0: int = 0) -> None:
    """Device-resident "column FFTs": the tensors hold a row-major ``[n][stride]`` array whose first ``batch`` columns
    are transformed along the rows' axis, in place (transform ``c`` = elements ``c + j*stride``).  ``stride`` and
    ``batch`` powers of two, ``batch <= stride``, ``n >= 64``; ``batch >= 16`` is always served, 8 columns for every n
    but 2^6 / 2^12 / 2^13, 4 columns for n = 2^10 / 2^20 only -- anything narrower raises :class:`PhastPanic` with code
    ``ERR_INVALID_ARG`` before anything runs (transpose and use :func:`fft_dit_batched`).  With ``twiddle_n`` the first pass multiplies element
    ``j`` of column ``c`` by ``W_twiddle_n^(j*(twiddle_col0 + c))`` on load (the inter-factor twiddle of a four-step
    split).  No reference counterpart."""
    dtype, sfx = planner._dtype, planner._sfx
    re, im = _Slice(reals, dtype, "reals"), _Slice(imags, dtype, "imags")
    if not _same_place(re, im):
        raise TypeError("fft_dit_strided needs device tensors")
    if re.len != im.len:
        _check(2)
    if re.len < n * stride:
        raise ValueError("the tensors must hold n*stride elements")
    if twiddle_n:
        _check(getattr(_lib.lib(), f"phast_fft_{sfx}_dit_strided_tw_dev")(
            re.ptr, im.ptr, C.c_size_t(n), C.c_size_t(batch), C.c_size_t(1), C.c_size_t(stride), C.c_int(int(direction)),
            planner._h, C.c_size_t(twiddle_n), C.c_size_t(twiddle_col0), _stream()))
        return
    _check(getattr(_lib.lib(), f"phast_fft_{sfx}_dit_strided_dev")(re.ptr, im.ptr, C.c_size_t(n), C.c_size_t(batch),
                                                                   C.c_size_t(1), C.c_size_t(stride),
                                                                   C.c_int(int(direction)), planner._h, _stream()))


def r2c_fft_batched(input_re, output_re, output_im, planner, batch: int) -> None:
    """Device-resident batch of R2C transforms: inputs ``n`` apart, outputs ``n/2 + 1`` apart."""
    fs = "f64" if planner._dtype == np.float64 else "f32"
    i, ore, oim = (_Slice(x, planner._dtype, w) for x, w in ((input_re, "input_re"), (output_re, "output_re"),
                                                              (output_im, "output_im")))
    n, out = planner.n, planner.n // 2 + 1
    if not _same_place(i, ore, oim) or i.len != batch * n or ore.len != batch * out or oim.len != batch * out:
        raise ValueError("need device tensors of batch*n, batch*(n/2+1), batch*(n/2+1) elements")
    _check(getattr(_lib.lib(), f"phast_r2c_fft_{fs}_dev")(i.ptr, ore.ptr, oim.ptr, C.c_size_t(batch), C.c_size_t(n),
                                                          C.c_size_t(out), planner._h, _stream()))


def c2r_fft_batched(input_re, input_im, output, planner, batch: int) -> None:
    """Device-resident batch of C2R transforms: inputs ``n/2 + 1`` apart, outputs ``n`` apart."""
    fs = "f64" if planner._dtype == np.float64 else "f32"
    ire, iim, out = (_Slice(x, planner._dtype, w) for x, w in ((input_re, "input_re"), (input_im, "input_im"),
                                                                (output, "output")))
    n, half1 = planner.n, planner.n // 2 + 1
    if not _same_place(ire, iim, out) or out.len != batch * n or ire.len != batch * half1 or iim.len != batch * half1:
        raise ValueError("need device tensors of batch*(n/2+1), batch*(n/2+1), batch*n elements")
    _check(getattr(_lib.lib(), f"phast_c2r_fft_{fs}_dev")(ire.ptr, iim.ptr, out.ptr, C.c_size_t(batch), C.c_size_t(half1),
                                                          C.c_size_t(n), planner._h, _stream()))


class TwiddleGrid64:
    """Device tables of W_N for the inter-factor twiddle of a four-step split (``include/phastft_hip.h``:
    ``phast_twiddle_grid64_*``): ``apply`` multiplies element (r, c) of a row-major device block by
    ``W_N^((row0 + r)*(col0 + c))`` in place.  Used by :mod:`phastft_amd.distributed`."""

    _sfx = "64"
    _dtype = np.float64

    def __init__(self, n: int):
        self._h = C.c_void_p()
        _check(getattr(_lib.lib(), f"phast_twiddle_grid{self._sfx}_new")(C.c_size_t(n), C.byref(self._h)))
        self.n = n

    def __del__(self):
        try:
            h = getattr(self, "_h", None)
            if h is not None and h.value:
                getattr(_lib.lib(), f"phast_twiddle_grid{self._sfx}_free")(h)
                h.value = None
        except Exception:  # interpreter shutdown: modules may already be gone
            pass

    def apply(self, reals, imags, rows: int, cols: int, row0: int = 0, col0: int = 0, row_pitch: int | None = None):
        re, im = _Slice(reals, self._dtype, "reals"), _Slice(imags, self._dtype, "imags")
        if not _same_place(re, im):
            raise TypeError("TwiddleGrid.apply needs device tensors")
        pitch = cols if row_pitch is None else row_pitch
        if re.len != im.len or (rows and re.len < (rows - 1) * pitch + cols):
            raise ValueError("block does not fit the tensors")
        _check(getattr(_lib.lib(), f"phast_twiddle_grid{self._sfx}_apply_dev")(
            self._h, re.ptr, im.ptr, C.c_size_t(rows), C.c_size_t(cols), C.c_size_t(pitch), C.c_size_t(row0),
            C.c_size_t(col0), _stream()))


class TwiddleGrid32(TwiddleGrid64):
    _sfx = "32"
    _dtype = np.float32


# ---------------------------------------------------------------------------------------------
# bit reversal  (algorithms/bravo.rs:303,317; public with feature bench-internals)
# ---------------------------------------------------------------------------------------------
def _bit_rev(fs, dtype, data, n):
    d = _Slice(data, dtype, "data")
    l = _lib.lib()
    if d.len != (1 << n):
        raise PhastPanic(16, "Data length must be 2^n")  # bravo.rs:228
    if d.dev:
        _check(getattr(l, f"phast_bit_rev_{fs}_dev")(d.ptr, C.c_uint(n), C.c_size_t(1), C.c_size_t(d.len), _stream()))
    else:
        _check(getattr(l, f"phast_bit_rev_{fs}")(d.ptr, C.c_size_t(d.len), C.c_uint(n)))


def bit_rev_bravo_f64(data, n: int) -> None:
    """bravo.rs:317"""
    _bit_rev("f64", np.float64, data, n)


def bit_rev_bravo_f32(data, n: int) -> None:
    """bravo.rs:303"""
    _bit_rev("f32", np.float32, data, n)


# ---------------------------------------------------------------------------------------------
# Complex<T> <-> planes  (complex_nums.rs:11-56; public with feature bench-internals)
# ---------------------------------------------------------------------------------------------
def _np_dtype(x):
    if _is_torch(x):
        import torch

        return np.float64 if x.dtype == torch.float64 else np.float32
    return np.float64 if x.dtype == np.float64 else np.float32


def _scalars(x):
    """a Complex<T> array as its 2 n scalars (`bytemuck::cast_slice`, complex_nums.rs:26,38); real arrays pass through"""
    if _is_torch(x):
        import torch

        return torch.view_as_real(x).reshape(-1) if x.is_complex() else x
    return x.view(np.float64 if x.dtype == np.complex128 else np.float32) if np.iscomplexobj(x) else x


def _like(x, n, dtype):
    if _is_torch(x):
        import torch

        return torch.empty(n, dtype=torch.float64 if dtype == np.float64 else torch.float32, device=x.device)
    return np.empty(n, dtype)


def deinterleave(data):
    """complex_nums.rs:11-17: ``[1, 2, 3, 4] -> ([1, 3], [2, 4])`` for any length (an odd last element is dropped, as
    `chunks_exact(2)` does).  A numpy array (host slice) or a torch cuda tensor (device, asynchronous on the current stream);
    returns two new arrays of the same kind."""
    data = _scalars(data)
    dtype = _np_dtype(data)
    fs = "f64" if dtype == np.float64 else "f32"
    d = _Slice(data, dtype, "input")
    a, b = _like(data, d.len // 2, dtype), _like(data, d.len // 2, dtype)
    sa, sb = _Slice(a, dtype, "out_a"), _Slice(b, dtype, "out_b")
    if d.dev:
        _check(getattr(_lib.lib(), f"phast_deinterleave_{fs}_dev")(d.ptr, C.c_size_t(d.len), sa.ptr, sb.ptr, _stream()))
    else:
        _check(getattr(_lib.lib(), f"phast_deinterleave_{fs}")(d.ptr, C.c_size_t(d.len), sa.ptr, C.c_size_t(sa.len), sb.ptr, C.c_size_t(sb.len)))
    return a, b


def deinterleave_complex64(signal):
    """complex_nums.rs:25-28: a `&[Complex<f64>]` (numpy complex128 / torch complex128) into (reals, imags)"""
    return deinterleave(signal)


def deinterleave_complex32(signal):
    """complex_nums.rs:37-40: a `&[Complex<f32>]` (numpy complex64 / torch complex64) into (reals, imags)"""
    return deinterleave(signal)


def combine_re_im(reals, imags):
    """complex_nums.rs:47-56: (reals, imags) -> one Complex<T> array; panics (PhastPanic) unless the lengths agree"""
    dtype = _np_dtype(reals)
    fs = "f64" if dtype == np.float64 else "f32"
    r, m = _Slice(reals, dtype, "reals"), _Slice(imags, dtype, "imags")
    dev = _same_place(r, m)
    if r.len != m.len:
        raise PhastPanic(2, "assertion `left == right` failed")  # complex_nums.rs:48
    out = _like(reals, 2 * r.len, dtype)
    so = _Slice(out, dtype, "out")
    if dev:
        _check(getattr(_lib.lib(), f"phast_combine_re_im_{fs}_dev")(r.ptr, m.ptr, C.c_size_t(r.len), so.ptr, _stream()))
    else:
        _check(getattr(_lib.lib(), f"phast_combine_re_im_{fs}")(r.ptr, C.c_size_t(r.len), m.ptr, C.c_size_t(m.len), so.ptr, C.c_size_t(so.len)))
    if _is_torch(out):
        import torch

        return torch.view_as_complex(out.reshape(-1, 2))
    return out.view(np.complex128 if dtype == np.float64 else np.complex64)


# ---------------------------------------------------------------------------------------------
# R2C / C2R  (algorithms/r2c.rs:521-895)
# ---------------------------------------------------------------------------------------------
def _r2c(fs, dtype, input_re, output_re, output_im, planner=None):
    i, ore, oim = _Slice(input_re, dtype, "input_re"), _Slice(output_re, dtype, "output_re"), _Slice(
        output_im, dtype, "output_im")
    l = _lib.lib()
    if _same_place(i, ore, oim):
        own = planner is None
        if own:
            planner = (PlannerR2c64 if fs == "f64" else PlannerR2c32)(i.len)  # r2c.rs:522
        n, half = planner.n, planner.n // 2
        if i.len != n:
            _check(5)
        if ore.len != half + 1:
            _check(6)
        if oim.len != half + 1:
            _check(7)
        _check(getattr(l, f"phast_r2c_fft_{fs}_dev")(i.ptr, ore.ptr, oim.ptr, C.c_size_t(1), C.c_size_t(n),
                                                     C.c_size_t(half + 1), planner._h, _stream()))
        if own:
            import torch

            torch.cuda.current_stream().synchronize()
        return
    args = [i.ptr, C.c_size_t(i.len), ore.ptr, C.c_size_t(ore.len), oim.ptr, C.c_size_t(oim.len)]
    if planner is None:
        _check(getattr(l, f"phast_r2c_fft_{fs}")(*args))
    else:
        _check(getattr(l, f"phast_r2c_fft_{fs}_with_planner")(*args, planner._h))


def r2c_fft_f64(input_re, output_re, output_im) -> None:
    """r2c.rs:521"""
    _r2c("f64", np.float64, input_re, output_re, output_im)


def r2c_fft_f32(input_re, output_re, output_im) -> None:
    """r2c.rs:598"""
    _r2c("f32", np.float32, input_re, output_re, output_im)


def r2c_fft_f64_with_planner(input_re, output_re, output_im, planner: PlannerR2c64) -> None:
    """r2c.rs:535"""
    _r2c("f64", np.float64, input_re, output_re, output_im, planner)


def r2c_fft_f32_with_planner(input_re, output_re, output_im, planner: PlannerR2c32) -> None:
    """r2c.rs:607"""
    _r2c("f32", np.float32, input_re, output_re, output_im, planner)


def _c2r(fs, dtype, input_re, input_im, output, planner=None, scratch=None):
    ire, iim, out = _Slice(input_re, dtype, "input_re"), _Slice(input_im, dtype, "input_im"), _Slice(
        output, dtype, "output")
    l = _lib.lib()
    sc = None
    if scratch is not None:
        sc = (_Slice(scratch[0], dtype, "scratch_re"), _Slice(scratch[1], dtype, "scratch_im"))
    if _same_place(ire, iim, out):
        own = planner is None
        if own:
            planner = (PlannerR2c64 if fs == "f64" else PlannerR2c32)(out.len)  # r2c.rs:696
        n, half = planner.n, planner.n // 2
        if out.len != n:
            _check(8)
        if ire.len != half + 1:
            _check(9)
        if iim.len != half + 1:
            _check(10)
        if sc is not None and sc[0].len != half:
            _check(11)
        if sc is not None and sc[1].len != half:
            _check(12)
        _check(getattr(l, f"phast_c2r_fft_{fs}_dev")(ire.ptr, iim.ptr, out.ptr, C.c_size_t(1), C.c_size_t(half + 1),
                                                     C.c_size_t(n), planner._h, _stream()))
        if own:
            import torch

            torch.cuda.current_stream().synchronize()
        return
    args = [ire.ptr, C.c_size_t(ire.len), iim.ptr, C.c_size_t(iim.len), out.ptr, C.c_size_t(out.len)]
    if planner is None:
        _check(getattr(l, f"phast_c2r_fft_{fs}")(*args))
    elif sc is None:
        _check(getattr(l, f"phast_c2r_fft_{fs}_with_planner")(*args, planner._h))
    else:
        _check(getattr(l, f"phast_c2r_fft_{fs}_with_planner_and_scratch")(
            *args, planner._h, sc[0].ptr, C.c_size_t(sc[0].len), sc[1].ptr, C.c_size_t(sc[1].len)))


def c2r_fft_f64(input_re, input_im, output) -> None:
    """r2c.rs:695"""
    _c2r("f64", np.float64, input_re, input_im, output)


def c2r_fft_f32(input_re, input_im, output) -> None:
    """r2c.rs:804"""
    _c2r("f32", np.float32, input_re, input_im, output)


def c2r_fft_f64_with_planner(input_re, input_im, output, planner: PlannerR2c64) -> None:
    """r2c.rs:710"""
    _c2r("f64", np.float64, input_re, input_im, output, planner)


def c2r_fft_f32_with_planner(input_re, input_im, output, planner: PlannerR2c32) -> None:
    """r2c.rs:813"""
    _c2r("f32", np.float32, input_re, input_im, output, planner)


def c2r_fft_f64_with_planner_and_scratch(input_re, input_im, output, planner, scratch_re, scratch_im) -> None:
    """r2c.rs:740"""
    _c2r("f64", np.float64, input_re, input_im, output, planner, (scratch_re, scratch_im))


def c2r_fft_f32_with_planner_and_scratch(input_re, input_im, output, planner, scratch_re, scratch_im) -> None:
    """r2c.rs:836"""
    _c2r("f32", np.float32, input_re, input_im, output, planner, (scratch_re, scratch_im))


# ---------------------------------------------------------------------------------------------
# harness helpers
# ---------------------------------------------------------------------------------------------
def fill_uniform(reals, imags, n: int, seed: int = 0xCAFE, first_id: int = 0) -> None:
    """Fill device tensors holding ``len/n`` transforms with the counter-based uniform [-1, 1) input
    (``imags`` may be None for real input)."""
    import torch

    dtype = np.float64 if reals.dtype == torch.float64 else np.float32
    fs = "f64" if dtype == np.float64 else "f32"
    re = _Slice(reals, dtype, "reals")
    im_ptr = _Slice(imags, dtype, "imags").ptr if imags is not None else C.c_void_p(0)
    _check(getattr(_lib.lib(), f"phast_fill_{fs}_dev")(re.ptr, im_ptr, C.c_size_t(n), C.c_size_t(re.len // n),
                                                       C.c_size_t(n), C.c_ulonglong(seed), C.c_ulonglong(first_id),
                                                       _stream()))


def digest(reals, imags, n: int, probe: int = 1):
    """Per-transform digest [sum re, sum im, sum |z|^2, re[probe]] as an f64 tensor of shape (batch, 4)."""
    import torch

    dtype = np.float64 if reals.dtype == torch.float64 else np.float32
    fs = "f64" if dtype == np.float64 else "f32"
    re, im = _Slice(reals, dtype, "reals"), _Slice(imags, dtype, "imags")
    batch = re.len // n
    out = torch.empty((batch, 4), dtype=torch.float64, device=reals.device)
    _check(getattr(_lib.lib(), f"phast_digest_{fs}_dev")(re.ptr, im.ptr, C.c_size_t(n), C.c_size_t(batch),
                                                         C.c_size_t(n), C.c_size_t(probe), C.c_void_p(out.data_ptr()),
                                                         _stream()))
    return out


def stream_probe(mib: int = 1024, reps: int = 5) -> dict:
    """This box's HBM streaming ceilings from the library's hand-written probe kernels (csrc/probe.hip): GB/s of a
    read-only, a write-only and a 1:1 copy kernel (read + write counted) over two buffers of ``mib`` MiB -- the figure a
    pass that reads and writes every byte once is to be read against."""
    import torch

    a = torch.empty(mib << 17, dtype=torch.float64, device="cuda").fill_(1.0)
    b = torch.empty_like(a)
    out = (C.c_double * 3)()
    _check(_lib.lib().phast_stream_probe_dev(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_size_t(mib << 20),
                                             C.c_int(reps), out, _stream()))
    del a, b
    return {"read": out[0], "write": out[1], "copy": out[2], "unit": "GB/s", "MiB": mib}


def debug_set_guard_bytes(nbytes: int) -> None:
    """debug: scratch buffers allocated from now on carry `nbytes` of 0xA5 guard band on either side"""
    _lib.lib().phast_debug_set_guard_bytes(C.c_size_t(nbytes))


def graph_upload(graph, stream=None) -> bool:
    """hipGraphUpload of an instantiated ``torch.cuda.CUDAGraph`` (so that its first replay does not pay the upload);
    False when this torch build does not expose the exec handle."""
    get = getattr(graph, "raw_cuda_graph_exec", None)
    if get is None:
        return False
    try:
        handle = get()
    except Exception:  # handle not kept by this torch version / graph not instantiated yet
        return False
    import torch

    s = stream if stream is not None else torch.cuda.current_stream()
    _check(_lib.lib().phast_hip_graph_upload(C.c_void_p(int(handle)), C.c_void_p(s.cuda_stream)))
    return True


def device_info() -> dict:
    name = C.create_string_buffer(256)
    cus, lds, mem = C.c_int(), C.c_size_t(), C.c_size_t()
    _check(_lib.lib().phast_device_info(name, C.c_size_t(256), C.byref(cus), C.byref(lds), C.byref(mem)))
    return {"name": name.value.decode(), "compute_units": cus.value, "lds_per_block": lds.value,
            "global_mem_bytes": mem.value}
