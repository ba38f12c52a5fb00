//! `phastft::algorithms` -- private in the reference unless feature `bench-internals` is on (lib.rs:20-23);
//! the same visibility here, so `phastft::algorithms::bravo::bit_rev_bravo_f64` resolves exactly when it
//! does upstream (benches/bit_reversal.rs:3).
pub mod bravo;
pub mod dit;
pub mod r2c;
