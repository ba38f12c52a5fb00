//! PhastFT's public API (planar slices) on AMD MI355X through `libphastft_hip.so`.
//!
//! Every item mirrors the item of the same name in PhastFT 0.3.0 (`src/lib.rs`, `src/planner.rs`,
//! `src/options.rs`, `src/algorithms/r2c.rs`): same signature, same in-place semantics, and the same
//! panics -- the C ABI returns one status per reference `assert!` and `phast_strerror` returns the
//! reference's panic text, which this shim re-raises with `panic!`.
//!
//! SOURCE ONLY: written against `include/phastft_hip.h`; never compiled (no Rust toolchain in the build image).
#![allow(clippy::missing_safety_doc)]

use std::ffi::{c_char, c_int, c_void, CStr};

#[repr(C)]
struct PhastOptions {
    multithreaded_bit_reversal: c_int,
    smallest_parallel_chunk_size: usize,
}

#[repr(C)]
struct Opaque {
    _private: [u8; 0],
}

extern "C" {
    fn phast_strerror(code: c_int) -> *const c_char;
    fn phast_last_hip_error() -> *const c_char;
    fn phast_options_guess(input_size: usize, out: *mut PhastOptions) -> c_int;
    fn phast_planner_dit64_with_mode(n: usize, mode: c_int, out: *mut *mut Opaque) -> c_int;
    fn phast_planner_dit32_with_mode(n: usize, mode: c_int, out: *mut *mut Opaque) -> c_int;
    fn phast_planner_dit64_free(p: *mut Opaque);
    fn phast_planner_dit32_free(p: *mut Opaque);
    fn phast_planner_r2c64_new(n: usize, out: *mut *mut Opaque) -> c_int;
    fn phast_planner_r2c32_new(n: usize, out: *mut *mut Opaque) -> c_int;
    fn phast_planner_r2c64_free(p: *mut Opaque);
    fn phast_planner_r2c32_free(p: *mut Opaque);
    fn phast_fft_64_dit_with_planner_and_opts(re: *mut f64, re_len: usize, im: *mut f64, im_len: usize,
        direction: c_int, planner: *const Opaque, opts: *const PhastOptions) -> c_int;
    fn phast_fft_32_dit_with_planner_and_opts(re: *mut f32, re_len: usize, im: *mut f32, im_len: usize,
        direction: c_int, planner: *const Opaque, opts: *const PhastOptions) -> c_int;
    fn phast_r2c_fft_f64_with_planner(input: *const f64, n: usize, ore: *mut f64, ore_len: usize, oim: *mut f64,
        oim_len: usize, planner: *const Opaque) -> c_int;
    fn phast_r2c_fft_f32_with_planner(input: *const f32, n: usize, ore: *mut f32, ore_len: usize, oim: *mut f32,
        oim_len: usize, planner: *const Opaque) -> c_int;
    fn phast_c2r_fft_f64_with_planner_and_scratch(ire: *const f64, ire_len: usize, iim: *const f64, iim_len: usize,
        out: *mut f64, out_len: usize, planner: *const Opaque, sre: *mut f64, sre_len: usize, sim: *mut f64,
        sim_len: usize) -> c_int;
    fn phast_c2r_fft_f32_with_planner_and_scratch(ire: *const f32, ire_len: usize, iim: *const f32, iim_len: usize,
        out: *mut f32, out_len: usize, planner: *const Opaque, sre: *mut f32, sre_len: usize, sim: *mut f32,
        sim_len: usize) -> c_int;
    fn phast_fft_64_dit_dev(re: *mut f64, im: *mut f64, n: usize, batch: usize, dist: usize, direction: c_int,
        planner: *const Opaque, stream: *mut c_void) -> c_int;
    #[cfg(feature = "complex-nums")]
    fn phast_fft_64_interleaved_with_planner_and_opts(signal: *mut f64, n: usize, direction: c_int,
        planner: *const Opaque, opts: *const PhastOptions) -> c_int;
    #[cfg(feature = "complex-nums")]
    fn phast_fft_32_interleaved_with_planner_and_opts(signal: *mut f32, n: usize, direction: c_int,
        planner: *const Opaque, opts: *const PhastOptions) -> c_int;
}

#[track_caller]
fn check(rc: c_int) {
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

/// planner.rs:10-16
#[derive(Copy, Clone)]
pub enum Direction {
    Forward = 1,
    Reverse = -1,
}

/// planner.rs:24-32 (`Tune` is accepted and ignored, as in the reference)
#[derive(Copy, Clone, Debug, Default)]
pub enum PlannerMode {
    #[default]
    Heuristic,
    Tune,
}

/// options.rs:8-43 -- CPU threading knobs, carried for source compatibility and ignored on the GPU
#[non_exhaustive]
#[derive(Debug, Clone)]
pub struct Options {
    pub multithreaded_bit_reversal: bool,
    pub smallest_parallel_chunk_size: usize,
}

impl Default for Options {
    fn default() -> Self {
        Self { multithreaded_bit_reversal: false, smallest_parallel_chunk_size: 16384 }
    }
}

impl Options {
    pub fn guess_options(input_size: usize) -> Options {
        let mut o = PhastOptions { multithreaded_bit_reversal: 0, smallest_parallel_chunk_size: 0 };
        check(unsafe { phast_options_guess(input_size, &mut o) });
        Options {
            multithreaded_bit_reversal: o.multithreaded_bit_reversal != 0,
            smallest_parallel_chunk_size: o.smallest_parallel_chunk_size,
        }
    }
    fn to_c(&self) -> PhastOptions {
        PhastOptions {
            multithreaded_bit_reversal: self.multithreaded_bit_reversal as c_int,
            smallest_parallel_chunk_size: self.smallest_parallel_chunk_size,
        }
    }
}

macro_rules! impl_planner_dit {
    ($name:ident, $new:ident, $free:ident) => {
        /// planner.rs:34-114: owns device twiddle tables and the device scratch buffer
        pub struct $name {
            h: *mut Opaque,
        }
        // the handle is immutable after creation; calls on one planner are serialised inside the library
        unsafe impl Send for $name {}
        unsafe impl Sync for $name {}
        impl $name {
            pub fn new(num_points: usize) -> Self {
                Self::with_mode(num_points, PlannerMode::Heuristic)
            }
            pub fn with_mode(num_points: usize, mode: PlannerMode) -> Self {
                let mut h = std::ptr::null_mut();
                check(unsafe { $new(num_points, mode as c_int, &mut h) });
                Self { h }
            }
        }
        impl Drop for $name {
            fn drop(&mut self) {
                unsafe { $free(self.h) }
            }
        }
    };
}
impl_planner_dit!(PlannerDit64, phast_planner_dit64_with_mode, phast_planner_dit64_free);
impl_planner_dit!(PlannerDit32, phast_planner_dit32_with_mode, phast_planner_dit32_free);

macro_rules! impl_planner_r2c {
    ($name:ident, $new:ident, $free:ident) => {
        /// planner.rs:164-212
        pub struct $name {
            h: *mut Opaque,
            n: usize,
        }
        unsafe impl Send for $name {}
        unsafe impl Sync for $name {}
        impl $name {
            pub fn new(n: usize) -> Self {
                let mut h = std::ptr::null_mut();
                check(unsafe { $new(n, &mut h) });
                Self { h, n }
            }
        }
        impl Drop for $name {
            fn drop(&mut self) {
                unsafe { $free(self.h) }
            }
        }
    };
}
impl_planner_r2c!(PlannerR2c64, phast_planner_r2c64_new, phast_planner_r2c64_free);
impl_planner_r2c!(PlannerR2c32, phast_planner_r2c32_new, phast_planner_r2c32_free);

macro_rules! impl_fft {
    ($t:ty, $planner:ident, $with_opts:ident, $with_planner:ident, $plain:ident, $c_fn:ident) => {
        /// algorithms/dit.rs:263 / 338
        pub fn $with_opts(reals: &mut [$t], imags: &mut [$t], direction: Direction, planner: &$planner, opts: &Options) {
            check(unsafe {
                $c_fn(reals.as_mut_ptr(), reals.len(), imags.as_mut_ptr(), imags.len(), direction as c_int, planner.h,
                      &opts.to_c())
            });
        }
        /// lib.rs:143 / 186
        pub fn $with_planner(reals: &mut [$t], imags: &mut [$t], direction: Direction, planner: &$planner) {
            let opts = Options::guess_options(reals.len());
            $with_opts(reals, imags, direction, planner, &opts);
        }
        /// lib.rs:180 / 223
        pub fn $plain(reals: &mut [$t], imags: &mut [$t], direction: Direction) {
            let planner = <$planner>::new(reals.len());
            $with_planner(reals, imags, direction, &planner);
        }
    };
}
impl_fft!(f64, PlannerDit64, fft_64_dit_with_planner_and_opts, fft_64_dit_with_planner, fft_64_dit,
          phast_fft_64_dit_with_planner_and_opts);
impl_fft!(f32, PlannerDit32, fft_32_dit_with_planner_and_opts, fft_32_dit_with_planner, fft_32_dit,
          phast_fft_32_dit_with_planner_and_opts);

macro_rules! impl_r2c {
    ($t:ty, $planner:ident, $r2c:ident, $r2c_p:ident, $c2r:ident, $c2r_p:ident, $c2r_ps:ident, $c_r2c:ident, $c_c2r:ident) => {
        /// r2c.rs:535 / 607
        pub fn $r2c_p(input_re: &[$t], output_re: &mut [$t], output_im: &mut [$t], planner: &$planner) {
            check(unsafe {
                $c_r2c(input_re.as_ptr(), input_re.len(), output_re.as_mut_ptr(), output_re.len(),
                       output_im.as_mut_ptr(), output_im.len(), planner.h)
            });
        }
        /// r2c.rs:521 / 598
        pub fn $r2c(input_re: &[$t], output_re: &mut [$t], output_im: &mut [$t]) {
            let planner = <$planner>::new(input_re.len());
            $r2c_p(input_re, output_re, output_im, &planner);
        }
        /// r2c.rs:740 / 836
        pub fn $c2r_ps(input_re: &[$t], input_im: &[$t], output: &mut [$t], planner: &$planner,
                       scratch_re: &mut [$t], scratch_im: &mut [$t]) {
            check(unsafe {
                $c_c2r(input_re.as_ptr(), input_re.len(), input_im.as_ptr(), input_im.len(), output.as_mut_ptr(),
                       output.len(), planner.h, scratch_re.as_mut_ptr(), scratch_re.len(), scratch_im.as_mut_ptr(),
                       scratch_im.len())
            });
        }
        /// r2c.rs:710 / 813
        pub fn $c2r_p(input_re: &[$t], input_im: &[$t], output: &mut [$t], planner: &$planner) {
            let half = planner.n / 2;
            let (mut sre, mut sim) = (vec![0.0 as $t; half], vec![0.0 as $t; half]);
            $c2r_ps(input_re, input_im, output, planner, &mut sre, &mut sim);
        }
        /// r2c.rs:695 / 804
        pub fn $c2r(input_re: &[$t], input_im: &[$t], output: &mut [$t]) {
            let planner = <$planner>::new(output.len());
            $c2r_p(input_re, input_im, output, &planner);
        }
    };
}
impl_r2c!(f64, PlannerR2c64, r2c_fft_f64, r2c_fft_f64_with_planner, c2r_fft_f64, c2r_fft_f64_with_planner,
          c2r_fft_f64_with_planner_and_scratch, phast_r2c_fft_f64_with_planner, phast_c2r_fft_f64_with_planner_and_scratch);
impl_r2c!(f32, PlannerR2c32, r2c_fft_f32, r2c_fft_f32_with_planner, c2r_fft_f32, c2r_fft_f32_with_planner,
          c2r_fft_f32_with_planner_and_scratch, phast_r2c_fft_f32_with_planner, phast_c2r_fft_f32_with_planner_and_scratch);

/// Device-resident, batched, asynchronous form (no reference counterpart): `d_reals`/`d_imags` are HIP device
/// pointers to `batch` transforms `dist` elements apart; `stream` is a `hipStream_t`.
pub unsafe fn fft_64_dit_dev(d_reals: *mut f64, d_imags: *mut f64, n: usize, batch: usize, dist: usize,
                             direction: Direction, planner: &PlannerDit64, stream: *mut c_void) {
    check(phast_fft_64_dit_dev(d_reals, d_imags, n, batch, dist, direction as c_int, planner.h, stream));
}

/// Interleaved `Complex<T>` signals (reference: feature `complex-nums`, lib.rs:41-140).  The reference copies into
/// two planar Vecs, runs the planar path and copies back; the library fuses the (de)interleave into the first
/// pass's load and the last pass's store, so there is no extra sweep.  `Complex<T>` is `repr(C)` (re, im).
#[cfg(feature = "complex-nums")]
pub use num_complex::Complex;

#[cfg(feature = "complex-nums")]
macro_rules! impl_interleaved {
    ($t:ty, $planner:ident, $with_opts:ident, $with_planner:ident, $plain:ident, $c_fn:ident) => {
        /// lib.rs:50
        pub fn $with_opts(signal: &mut [Complex<$t>], direction: Direction, planner: &$planner, opts: &Options) {
            let c_opts = opts.to_c();
            check(unsafe {
                $c_fn(signal.as_mut_ptr() as *mut $t, signal.len(), direction as c_int, planner.h, &c_opts)
            });
        }
        /// lib.rs:87
        pub fn $with_planner(signal: &mut [Complex<$t>], direction: Direction, planner: &$planner) {
            let opts = Options::guess_options(signal.len());
            $with_opts(signal, direction, planner, &opts);
        }
        /// lib.rs:120
        pub fn $plain(signal: &mut [Complex<$t>], direction: Direction) {
            let planner = <$planner>::new(signal.len());
            $with_planner(signal, direction, &planner);
        }
    };
}
#[cfg(feature = "complex-nums")]
impl_interleaved!(f64, PlannerDit64, fft_64_interleaved_with_planner_and_opts, fft_64_interleaved_with_planner,
                  fft_64_interleaved, phast_fft_64_interleaved_with_planner_and_opts);
#[cfg(feature = "complex-nums")]
impl_interleaved!(f32, PlannerDit32, fft_32_interleaved_with_planner_and_opts, fft_32_interleaved_with_planner,
                  fft_32_interleaved, phast_fft_32_interleaved_with_planner_and_opts);
