//! PhastFT's public API (planar slices) on AMD MI355X through `libphastft_hip.so`.
//!
//! The module tree is the reference's (PhastFT 0.3.0 `src/lib.rs:20-38`): `phastft::planner::{Direction,
//! PlannerMode, PlannerDit64, PlannerDit32, PlannerR2c64, PlannerR2c32}`, `phastft::options::Options`, the
//! transforms at the crate root, and -- behind feature `bench-internals`, as upstream --
//! `phastft::algorithms::bravo::{bit_rev_bravo_f64, bit_rev_bravo_f32}`.  Depend on it under the reference's
//! name and existing call sites compile unchanged:
//!
//! ```toml
//! phastft = { package = "phastft-hip", path = "rust/phastft-hip" }
//! ```
//!
//! Every item has the signature, in-place semantics and panics of the item of the same name upstream: the C
//! ABI returns one status per reference `assert!`, `phast_strerror` returns the reference's panic text, and
//! `ffi::check` re-raises it with `panic!`.
//!
//! SOURCE ONLY: written against `include/phastft_hip.h`; the build image has no Rust toolchain, so
//! `tests/test_rust_shim.py` checks the `extern "C"` block against the header instead of `cargo build`.
#![allow(clippy::missing_safety_doc)]

#[cfg(not(feature = "bench-internals"))]
mod algorithms;
#[cfg(feature = "bench-internals")]
pub mod algorithms;
// lib.rs:23-27 of the reference: complex_nums is private with `complex-nums` alone, public with `bench-internals`
#[cfg(all(feature = "complex-nums", not(feature = "bench-internals")))]
mod complex_nums;
#[cfg(feature = "bench-internals")]
pub mod complex_nums;
mod ffi;
pub mod options;
pub mod planner;

pub use algorithms::dit::{fft_32_dit_with_planner_and_opts, fft_64_dit_with_planner_and_opts};
pub use algorithms::r2c::{
    c2r_fft_f32, c2r_fft_f32_with_planner, c2r_fft_f32_with_planner_and_scratch, c2r_fft_f64,
    c2r_fft_f64_with_planner, c2r_fft_f64_with_planner_and_scratch, r2c_fft_f32,
    r2c_fft_f32_with_planner, r2c_fft_f64, r2c_fft_f64_with_planner,
};

use crate::options::Options;
use crate::planner::{Direction, PlannerDit32, PlannerDit64};
use std::ffi::{c_int, c_void};

macro_rules! impl_fft {
    ($t:ty, $planner:ident, $with_opts:ident, $with_planner:ident, $plain:ident) => {
        /// lib.rs:143 / 186
        pub fn $with_planner(reals: &mut [$t], imags: &mut [$t], direction: Direction, planner: &$planner) {
            let opts = Options::guess_options(reals.len());
            $with_opts(reals, imags, direction, planner, &opts);
        }
        /// lib.rs:180 / 223 -- the planner is built from `reals.len()` first, so a bad length panics there
        pub fn $plain(reals: &mut [$t], imags: &mut [$t], direction: Direction) {
            let planner = <$planner>::new(reals.len());
            $with_planner(reals, imags, direction, &planner);
        }
    };
}
impl_fft!(f64, PlannerDit64, fft_64_dit_with_planner_and_opts, fft_64_dit_with_planner, fft_64_dit);
impl_fft!(f32, PlannerDit32, fft_32_dit_with_planner_and_opts, fft_32_dit_with_planner, fft_32_dit);

/// Device-resident, batched, asynchronous forms (no reference counterpart): `d_reals`/`d_imags` are HIP device
/// pointers to `batch` transforms `dist` elements apart; `stream` is a `hipStream_t`.
pub unsafe fn fft_64_dit_dev(d_reals: *mut f64, d_imags: *mut f64, n: usize, batch: usize, dist: usize,
                             direction: Direction, planner: &PlannerDit64, stream: *mut c_void) {
    ffi::check(ffi::phast_fft_64_dit_dev(d_reals, d_imags, n, batch, dist, direction as c_int, planner.h, stream));
}
/// f32 twin of [`fft_64_dit_dev`]
pub unsafe fn fft_32_dit_dev(d_reals: *mut f32, d_imags: *mut f32, n: usize, batch: usize, dist: usize,
                             direction: Direction, planner: &PlannerDit32, stream: *mut c_void) {
    ffi::check(ffi::phast_fft_32_dit_dev(d_reals, d_imags, n, batch, dist, direction as c_int, planner.h, stream));
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
            ffi::check(unsafe {
                ffi::$c_fn(signal.as_mut_ptr() as *mut $t, signal.len(), direction as c_int, planner.h, &c_opts)
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
