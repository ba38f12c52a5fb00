//! `phastft::planner` (planner.rs:10-212): `Direction`, `PlannerMode` and the four planners.  A planner owns
//! device twiddle tables and device scratch inside `libphastft_hip.so`; like the reference's planners it is an
//! immutable value any number of callers may borrow (`Send + Sync`, planner.rs:38-39) -- inside the library every
//! concurrent caller works in a workspace of its own (scratch, staging, stream), so borrowed planners run side by side.
use crate::ffi::{self, Opaque};
use std::ffi::c_int;

/// planner.rs:10-16
#[derive(Copy, Clone)]
pub enum Direction {
    Forward = 1,
    Reverse = -1,
}

/// planner.rs:24-32.  `Tune`: the library times the plans that exist for this length on the device at plan time and keeps
/// the fastest (include/phastft_hip.h, "PlannerMode::Tune"); `Heuristic`: static rules, zero planning overhead.
#[derive(Copy, Clone, Debug, Default)]
pub enum PlannerMode {
    #[default]
    Heuristic,
    Tune,
}

/// PHAST_TUNE_* (include/phastft_hip.h): which call a tuning run measures
#[derive(Copy, Clone, Debug)]
pub enum TuneKind {
    C2C = 0,
    C2CInterleaved = 1,
    R2C = 2,
    C2R = 3,
}

/// What tuning runs found, as text (one line per type / kind / length / batch bucket): carry it to the next process, or set
/// `PHAST_WISDOM=<path>` and the library does.
pub fn wisdom_export() -> String {
    let mut need = 0usize;
    ffi::check(unsafe { ffi::phast_wisdom_export(std::ptr::null_mut(), 0, &mut need) });
    let mut buf = vec![0u8; need];
    ffi::check(unsafe { ffi::phast_wisdom_export(buf.as_mut_ptr() as *mut _, need, std::ptr::null_mut()) });
    buf.pop(); // the NUL
    String::from_utf8(buf).expect("wisdom is ASCII")
}
pub fn wisdom_import(text: &str) {
    let c = std::ffi::CString::new(text).expect("no NUL in wisdom text");
    ffi::check(unsafe { ffi::phast_wisdom_import(c.as_ptr()) });
}
pub fn wisdom_forget() {
    unsafe { ffi::phast_wisdom_forget() }
}

macro_rules! impl_planner_dit {
    ($name:ident, $new:ident, $free:ident, $tune:ident) => {
        /// planner.rs:34-114
        pub struct $name {
            pub(crate) h: *mut Opaque,
        }
        // SAFETY: the handle is immutable after creation; what a call mutates lives in a per-call workspace inside the library
        unsafe impl Send for $name {}
        unsafe impl Sync for $name {}
        impl $name {
            /// planner.rs:55 -- panics unless `num_points` is a power of two > 0 (planner.rs:66)
            pub fn new(num_points: usize) -> Self {
                Self::with_mode(num_points, PlannerMode::Heuristic)
            }
            /// planner.rs:65
            pub fn with_mode(num_points: usize, mode: PlannerMode) -> Self {
                let mut h = std::ptr::null_mut();
                ffi::check(unsafe { ffi::$new(num_points, mode as c_int, &mut h) });
                Self { h }
            }
            /// No reference counterpart: `PlannerMode::Tune` for `batch` transforms per call of `kind` (the reference's
            /// planners only ever see one transform per call, which `with_mode(_, Tune)` covers).
            pub fn tune(&mut self, batch: usize, kind: TuneKind) -> ffi::PhastTuneReport {
                let mut rep = std::mem::MaybeUninit::<ffi::PhastTuneReport>::zeroed();
                ffi::check(unsafe { ffi::$tune(self.h, batch, kind as c_int, rep.as_mut_ptr()) });
                unsafe { rep.assume_init() }
            }
        }
        impl Drop for $name {
            fn drop(&mut self) {
                unsafe { ffi::$free(self.h) }
            }
        }
    };
}
impl_planner_dit!(PlannerDit64, phast_planner_dit64_with_mode, phast_planner_dit64_free, phast_planner_dit64_tune);
impl_planner_dit!(PlannerDit32, phast_planner_dit32_with_mode, phast_planner_dit32_free, phast_planner_dit32_tune);

macro_rules! impl_planner_r2c {
    ($name:ident, $new:ident, $free:ident, $tune:ident) => {
        /// planner.rs:164-212 -- one planner drives both R2C and C2R (planner.rs:171-172)
        pub struct $name {
            pub(crate) h: *mut Opaque,
            pub(crate) n: usize,
        }
        // SAFETY: as for the DIT planners
        unsafe impl Send for $name {}
        unsafe impl Sync for $name {}
        impl $name {
            /// planner.rs:194 -- panics with "n must be a power of 2 >= 4" (planner.rs:195)
            pub fn new(n: usize) -> Self {
                Self::with_mode(n, PlannerMode::Heuristic)
            }
            /// No reference counterpart (PlannerR2c*::new has no mode): the switch of `PlannerDit*::with_mode` for
            /// `r2c_fft_*` and `c2r_fft_*`.
            pub fn with_mode(n: usize, mode: PlannerMode) -> Self {
                let mut h = std::ptr::null_mut();
                ffi::check(unsafe { ffi::$new(n, mode as c_int, &mut h) });
                Self { h, n }
            }
            pub fn tune(&mut self, batch: usize, kind: TuneKind) -> ffi::PhastTuneReport {
                let mut rep = std::mem::MaybeUninit::<ffi::PhastTuneReport>::zeroed();
                ffi::check(unsafe { ffi::$tune(self.h, batch, kind as c_int, rep.as_mut_ptr()) });
                unsafe { rep.assume_init() }
            }
        }
        impl Drop for $name {
            fn drop(&mut self) {
                unsafe { ffi::$free(self.h) }
            }
        }
    };
}
impl_planner_r2c!(PlannerR2c64, phast_planner_r2c64_with_mode, phast_planner_r2c64_free, phast_planner_r2c64_tune);
impl_planner_r2c!(PlannerR2c32, phast_planner_r2c32_with_mode, phast_planner_r2c32_free, phast_planner_r2c32_tune);
