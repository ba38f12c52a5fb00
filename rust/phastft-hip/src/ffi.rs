//! Raw bindings to `libphastft_hip.so` -- one declaration per entry point of `include/phastft_hip.h` that the
//! safe wrappers use.  `tests/test_rust_shim.py` parses this block and checks every symbol and its arity
//! against the C header (no Rust toolchain in the build image).
use std::ffi::{c_char, c_int, c_uint, c_void, CStr};

#[repr(C)]
pub(crate) struct PhastOptions {
    pub multithreaded_bit_reversal: c_int,
    pub smallest_parallel_chunk_size: usize,
}

/// `phast_tune_report` (include/phastft_hip.h: PlannerMode::Tune)
#[repr(C)]
pub struct PhastTuneReport {
    pub adopted: c_int,
    pub candidates: c_uint,
    pub us_heuristic: f32,
    pub us_best: f32,
    pub seconds: f64,
    pub plan: [c_char; 96],
}

#[repr(C)]
pub(crate) struct Opaque {
    _private: [u8; 0],
}

extern "C" {
    pub(crate) fn phast_strerror(code: c_int) -> *const c_char;
    pub(crate) fn phast_last_hip_error() -> *const c_char;
    pub(crate) fn phast_options_guess(input_size: usize, out: *mut PhastOptions) -> c_int;
    pub(crate) fn phast_planner_dit64_with_mode(n: usize, mode: c_int, out: *mut *mut Opaque) -> c_int;
    pub(crate) fn phast_planner_dit32_with_mode(n: usize, mode: c_int, out: *mut *mut Opaque) -> c_int;
    pub(crate) fn phast_planner_dit64_free(p: *mut Opaque);
    pub(crate) fn phast_planner_dit32_free(p: *mut Opaque);
    pub(crate) fn phast_planner_r2c64_with_mode(n: usize, mode: c_int, out: *mut *mut Opaque) -> c_int;
    pub(crate) fn phast_planner_r2c32_with_mode(n: usize, mode: c_int, out: *mut *mut Opaque) -> c_int;
    pub(crate) fn phast_planner_dit64_tune(p: *mut Opaque, batch_hint: usize, kind: c_int, report: *mut PhastTuneReport) -> c_int;
    pub(crate) fn phast_planner_dit32_tune(p: *mut Opaque, batch_hint: usize, kind: c_int, report: *mut PhastTuneReport) -> c_int;
    pub(crate) fn phast_planner_r2c64_tune(p: *mut Opaque, batch_hint: usize, kind: c_int, report: *mut PhastTuneReport) -> c_int;
    pub(crate) fn phast_planner_r2c32_tune(p: *mut Opaque, batch_hint: usize, kind: c_int, report: *mut PhastTuneReport) -> c_int;
    pub(crate) fn phast_wisdom_export(buf: *mut c_char, buf_len: usize, needed: *mut usize) -> c_int;
    pub(crate) fn phast_wisdom_import(text: *const c_char) -> c_int;
    pub(crate) fn phast_wisdom_forget();
    pub(crate) fn phast_planner_r2c64_free(p: *mut Opaque);
    pub(crate) fn phast_planner_r2c32_free(p: *mut Opaque);
    pub(crate) fn phast_fft_64_dit_with_planner_and_opts(re: *mut f64, re_len: usize, im: *mut f64, im_len: usize,
        direction: c_int, planner: *const Opaque, opts: *const PhastOptions) -> c_int;
    pub(crate) fn phast_fft_32_dit_with_planner_and_opts(re: *mut f32, re_len: usize, im: *mut f32, im_len: usize,
        direction: c_int, planner: *const Opaque, opts: *const PhastOptions) -> c_int;
    pub(crate) fn phast_r2c_fft_f64_with_planner(input: *const f64, n: usize, ore: *mut f64, ore_len: usize,
        oim: *mut f64, oim_len: usize, planner: *const Opaque) -> c_int;
    pub(crate) fn phast_r2c_fft_f32_with_planner(input: *const f32, n: usize, ore: *mut f32, ore_len: usize,
        oim: *mut f32, oim_len: usize, planner: *const Opaque) -> c_int;
    pub(crate) fn phast_c2r_fft_f64_with_planner_and_scratch(ire: *const f64, ire_len: usize, iim: *const f64,
        iim_len: usize, out: *mut f64, out_len: usize, planner: *const Opaque, sre: *mut f64, sre_len: usize,
        sim: *mut f64, sim_len: usize) -> c_int;
    pub(crate) fn phast_c2r_fft_f32_with_planner_and_scratch(ire: *const f32, ire_len: usize, iim: *const f32,
        iim_len: usize, out: *mut f32, out_len: usize, planner: *const Opaque, sre: *mut f32, sre_len: usize,
        sim: *mut f32, sim_len: usize) -> c_int;
    pub(crate) fn phast_bit_rev_f64(data: *mut f64, len: usize, log_n: c_uint) -> c_int;
    pub(crate) fn phast_bit_rev_f32(data: *mut f32, len: usize, log_n: c_uint) -> c_int;
    pub(crate) fn phast_deinterleave_f64(input: *const f64, len: usize, a: *mut f64, a_len: usize, b: *mut f64,
        b_len: usize) -> c_int;
    pub(crate) fn phast_deinterleave_f32(input: *const f32, len: usize, a: *mut f32, a_len: usize, b: *mut f32,
        b_len: usize) -> c_int;
    pub(crate) fn phast_combine_re_im_f64(re: *const f64, re_len: usize, im: *const f64, im_len: usize, out: *mut f64,
        out_len: usize) -> c_int;
    pub(crate) fn phast_combine_re_im_f32(re: *const f32, re_len: usize, im: *const f32, im_len: usize, out: *mut f32,
        out_len: usize) -> c_int;
    pub(crate) fn phast_fft_64_dit_dev(re: *mut f64, im: *mut f64, n: usize, batch: usize, dist: usize,
        direction: c_int, planner: *const Opaque, stream: *mut c_void) -> c_int;
    pub(crate) fn phast_fft_32_dit_dev(re: *mut f32, im: *mut f32, n: usize, batch: usize, dist: usize,
        direction: c_int, planner: *const Opaque, stream: *mut c_void) -> c_int;
    pub(crate) fn phast_fft_64_interleaved_with_planner_and_opts(signal: *mut f64, n: usize, direction: c_int,
        planner: *const Opaque, opts: *const PhastOptions) -> c_int;
    pub(crate) fn phast_fft_32_interleaved_with_planner_and_opts(signal: *mut f32, n: usize, direction: c_int,
        planner: *const Opaque, opts: *const PhastOptions) -> c_int;
}

/// Re-raises a library status as the reference's panic: `phast_strerror` returns the exact text of the
/// `assert!` / `assert_eq!` the status stands for (include/phastft_hip.h, enum phast_status).
#[track_caller]
pub(crate) fn check(rc: c_int) {
    if rc != 0 {
        // SAFETY: both functions return static / thread-local NUL-terminated strings
        let msg = unsafe { CStr::from_ptr(phast_strerror(rc)) }.to_string_lossy();
        if rc >= 13 {
            let hip = unsafe { CStr::from_ptr(phast_last_hip_error()) }.to_string_lossy();
            panic!("{msg}: {hip}");
        }
        panic!("{msg}");
    }
}
