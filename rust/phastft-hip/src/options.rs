//! `phastft::options` (options.rs:8-43): the two CPU threading knobs, carried for source compatibility and
//! ignored on the GPU.
use crate::ffi;
use std::ffi::c_int;

/// options.rs:8-31
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
    /// options.rs:38-43 (`multithreaded_bit_reversal = log2(n) >= 16`; panics for 0 like `usize::ilog2`)
    pub fn guess_options(input_size: usize) -> Options {
        let mut o = ffi::PhastOptions { multithreaded_bit_reversal: 0, smallest_parallel_chunk_size: 0 };
        ffi::check(unsafe { ffi::phast_options_guess(input_size, &mut o) });
        Options {
            multithreaded_bit_reversal: o.multithreaded_bit_reversal != 0,
            smallest_parallel_chunk_size: o.smallest_parallel_chunk_size,
        }
    }
    pub(crate) fn to_c(&self) -> ffi::PhastOptions {
        ffi::PhastOptions {
            multithreaded_bit_reversal: self.multithreaded_bit_reversal as c_int,
            smallest_parallel_chunk_size: self.smallest_parallel_chunk_size,
        }
    }
}
