//! `phastft::complex_nums` (complex_nums.rs:11-56) -- public with feature `bench-internals`, as upstream (lib.rs:23-27):
//! `Complex<T>` arrays to planes and back.  The sweeps run on the GPU through the host-slice forms of the C ABI
//! (`phast_deinterleave_*`, `phast_combine_re_im_*`); the interleaved transforms at the crate root do not call them --
//! the library reads and writes pairs in its first and last pass.
use crate::ffi;
use num_complex::Complex;

/// complex_nums.rs:11 -- `[1, 2, 3, 4]` into `([1, 3], [2, 4])` for any length (an odd last element is dropped, as
/// `chunks_exact(2)` does).  The reference is generic over `T: Copy`; the library moves 4- and 8-byte scalars.
pub trait Scalar: Copy + Default {
    #[doc(hidden)]
    unsafe fn deinterleave_raw(input: *const Self, len: usize, a: *mut Self, b: *mut Self) -> std::ffi::c_int;
    #[doc(hidden)]
    unsafe fn combine_raw(re: *const Self, im: *const Self, n: usize, out: *mut Self) -> std::ffi::c_int;
}
impl Scalar for f64 {
    unsafe fn deinterleave_raw(input: *const f64, len: usize, a: *mut f64, b: *mut f64) -> std::ffi::c_int {
        ffi::phast_deinterleave_f64(input, len, a, len / 2, b, len / 2)
    }
    unsafe fn combine_raw(re: *const f64, im: *const f64, n: usize, out: *mut f64) -> std::ffi::c_int {
        ffi::phast_combine_re_im_f64(re, n, im, n, out, 2 * n)
    }
}
impl Scalar for f32 {
    unsafe fn deinterleave_raw(input: *const f32, len: usize, a: *mut f32, b: *mut f32) -> std::ffi::c_int {
        ffi::phast_deinterleave_f32(input, len, a, len / 2, b, len / 2)
    }
    unsafe fn combine_raw(re: *const f32, im: *const f32, n: usize, out: *mut f32) -> std::ffi::c_int {
        ffi::phast_combine_re_im_f32(re, n, im, n, out, 2 * n)
    }
}

/// complex_nums.rs:11
pub fn deinterleave<T: Scalar>(input: &[T]) -> (Vec<T>, Vec<T>) {
    let half = input.len() / 2;
    let (mut a, mut b) = (vec![T::default(); half], vec![T::default(); half]);
    ffi::check(unsafe { T::deinterleave_raw(input.as_ptr(), input.len(), a.as_mut_ptr(), b.as_mut_ptr()) });
    (a, b)
}

/// complex_nums.rs:25 -- `Complex<f64>` is `repr(C)` (re, im): the cast slice of the reference
pub fn deinterleave_complex64(signal: &[Complex<f64>]) -> (Vec<f64>, Vec<f64>) {
    deinterleave(unsafe { std::slice::from_raw_parts(signal.as_ptr() as *const f64, 2 * signal.len()) })
}

/// complex_nums.rs:37
pub fn deinterleave_complex32(signal: &[Complex<f32>]) -> (Vec<f32>, Vec<f32>) {
    deinterleave(unsafe { std::slice::from_raw_parts(signal.as_ptr() as *const f32, 2 * signal.len()) })
}

/// complex_nums.rs:47 -- panics if `reals.len() != imags.len()`
pub fn combine_re_im<T: Scalar>(reals: &[T], imags: &[T]) -> Vec<Complex<T>> {
    assert_eq!(reals.len(), imags.len());
    let mut out = vec![Complex { re: T::default(), im: T::default() }; reals.len()];
    ffi::check(unsafe { T::combine_raw(reals.as_ptr(), imags.as_ptr(), reals.len(), out.as_mut_ptr() as *mut T) });
    out
}
