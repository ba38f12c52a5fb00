//! `phastft::algorithms::dit` (algorithms/dit.rs:263-401): the planar in-place transform with an explicit
//! planner and options.  Re-exported at the crate root like the reference does (lib.rs:33).
use crate::ffi;
use crate::options::Options;
use crate::planner::{Direction, PlannerDit32, PlannerDit64};
use std::ffi::c_int;

macro_rules! impl_with_opts {
    ($t:ty, $planner:ident, $name:ident, $c_fn:ident) => {
        /// algorithms/dit.rs:263 / 338 -- panics like the reference: length mismatch (dit.rs:284), not a power
        /// of two (dit.rs:285), planner of another size (dit.rs:289)
        pub fn $name(reals: &mut [$t], imags: &mut [$t], direction: Direction, planner: &$planner, opts: &Options) {
            let c_opts = opts.to_c();
            ffi::check(unsafe {
                ffi::$c_fn(reals.as_mut_ptr(), reals.len(), imags.as_mut_ptr(), imags.len(), direction as c_int,
                           planner.h, &c_opts)
            });
        }
    };
}
impl_with_opts!(f64, PlannerDit64, fft_64_dit_with_planner_and_opts, phast_fft_64_dit_with_planner_and_opts);
impl_with_opts!(f32, PlannerDit32, fft_32_dit_with_planner_and_opts, phast_fft_32_dit_with_planner_and_opts);
