//! `phastft::algorithms::r2c` (algorithms/r2c.rs:521-895): real-to-complex and complex-to-real transforms.
//! Re-exported at the crate root like the reference does (lib.rs:34-38).
use crate::ffi;
use crate::planner::{PlannerR2c32, PlannerR2c64};

macro_rules! impl_r2c {
    ($t:ty, $planner:ident, $r2c:ident, $r2c_p:ident, $c2r:ident, $c2r_p:ident, $c2r_ps:ident, $c_r2c:ident, $c_c2r:ident) => {
        /// r2c.rs:535 / 607 -- the three length panics of r2c.rs:543-553 carry the reference's messages
        pub fn $r2c_p(input_re: &[$t], output_re: &mut [$t], output_im: &mut [$t], planner: &$planner) {
            ffi::check(unsafe {
                ffi::$c_r2c(input_re.as_ptr(), input_re.len(), output_re.as_mut_ptr(), output_re.len(),
                            output_im.as_mut_ptr(), output_im.len(), planner.h)
            });
        }
        /// r2c.rs:521 / 598
        pub fn $r2c(input_re: &[$t], output_re: &mut [$t], output_im: &mut [$t]) {
            let planner = <$planner>::new(input_re.len());
            $r2c_p(input_re, output_re, output_im, &planner);
        }
        /// r2c.rs:740 / 836 -- the scratch slices are length-checked (r2c.rs:760-769) and otherwise unused: the
        /// workspace lives in device memory
        pub fn $c2r_ps(input_re: &[$t], input_im: &[$t], output: &mut [$t], planner: &$planner,
                       scratch_re: &mut [$t], scratch_im: &mut [$t]) {
            ffi::check(unsafe {
                ffi::$c_c2r(input_re.as_ptr(), input_re.len(), input_im.as_ptr(), input_im.len(),
                            output.as_mut_ptr(), output.len(), planner.h, scratch_re.as_mut_ptr(), scratch_re.len(),
                            scratch_im.as_mut_ptr(), scratch_im.len())
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
          c2r_fft_f64_with_planner_and_scratch, phast_r2c_fft_f64_with_planner,
          phast_c2r_fft_f64_with_planner_and_scratch);
impl_r2c!(f32, PlannerR2c32, r2c_fft_f32, r2c_fft_f32_with_planner, c2r_fft_f32, c2r_fft_f32_with_planner,
          c2r_fft_f32_with_planner_and_scratch, phast_r2c_fft_f32_with_planner,
          phast_c2r_fft_f32_with_planner_and_scratch);
