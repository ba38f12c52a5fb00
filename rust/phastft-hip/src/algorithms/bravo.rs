//! `phastft::algorithms::bravo` (bravo.rs:303-324): in-place bit-reversal permutation of `2^n` elements.
//! The reference's first argument is a `fearless_simd` dispatch token; it only selects a CPU vector width,
//! so any value is accepted here and ignored (`dispatch!(level, simd => bit_rev_bravo_f64(simd, ..))` call
//! sites compile unchanged).
use crate::ffi;
use std::ffi::c_uint;

/// bravo.rs:317 -- panics unless `data.len() == 2^n` (bravo.rs:228)
pub fn bit_rev_bravo_f64<S>(_simd: S, data: &mut [f64], n: usize) {
    ffi::check(unsafe { ffi::phast_bit_rev_f64(data.as_mut_ptr(), data.len(), n as c_uint) });
}

/// bravo.rs:303
pub fn bit_rev_bravo_f32<S>(_simd: S, data: &mut [f32], n: usize) {
    ffi::check(unsafe { ffi::phast_bit_rev_f32(data.as_mut_ptr(), data.len(), n as c_uint) });
}
