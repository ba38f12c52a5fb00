"""The Rust shim crate (rust/phastft-hip) cannot be compiled in this image (no cargo/rustc), so its contract with
the C ABI is checked textually: every `extern "C"` declaration in src/ffi.rs must name a prototype of
include/phastft_hip.h with the same arity and the same argument / return type classes, and the crate must expose
the reference's module paths (PhastFT 0.3.0 src/lib.rs:20-38) so that upstream call sites compile unchanged:

    use phastft::planner::{Direction, PlannerDit32, PlannerDit64};      (examples/benchmark.rs:4)
    use phastft::options::Options;                                     (benches/bench.rs:13)
    use phastft::{fft_32_dit_with_planner_and_opts, ...};              (benches/bench.rs:15)
    use phastft::algorithms::bravo::{bit_rev_bravo_f32, bit_rev_bravo_f64};   (benches/bit_reversal.rs:3)
"""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "rust", "phastft-hip")


def _strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def c_prototypes():
    """name -> (return class, [argument classes]) from include/phastft_hip.h"""
    text = _strip_comments(open(os.path.join(ROOT, "include", "phastft_hip.h")).read())
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(phast_[a-z0-9_]+)\s*\(([^;{}]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret:
            continue
        arglist = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        protos[name] = (_c_class(ret), [_c_class(a) for a in arglist])
    return protos


def _c_class(decl: str) -> str:
    decl = decl.strip()
    stars = decl.count("*")
    base = decl.replace("*", " ")
    base = re.sub(r"\bconst\b", " ", base)
    toks = base.split()
    # drop the parameter name when there is one (the last token, unless the declaration is a bare type)
    types = {"int", "unsigned", "size_t", "double", "float", "void", "char", "long"}
    if len(toks) > 1 and toks[-1] not in types:
        toks = toks[:-1]
    base = " ".join(toks)
    if stars:
        if base in ("double", "float", "char", "void", "unsigned", "int", "size_t", "unsigned long long"):
            return f"ptr{stars}:{base}"
        return f"ptr{stars}:opaque"  # struct handles (phast_planner_*, phast_options)
    return {"int": "i32", "unsigned": "u32", "size_t": "usize", "void": "void", "double": "f64", "float": "f32",
            "unsigned long long": "u64"}[base]


def _rust_class(ty: str) -> str:
    ty = ty.strip()
    stars = 0
    while ty.startswith("*"):
        ty = re.sub(r"^\*(const|mut)\s+", "", ty)
        stars += 1
    prim = {"c_int": "i32", "c_uint": "u32", "usize": "usize", "f64": "f64", "f32": "f32"}
    if stars:
        base = {"f64": "double", "f32": "float", "c_char": "char", "c_void": "void", "c_uint": "unsigned", "usize": "size_t"}.get(ty)
        return f"ptr{stars}:{base}" if base else f"ptr{stars}:opaque"
    return prim[ty]


def rust_externs():
    text = _strip_comments(open(os.path.join(CRATE, "src", "ffi.rs")).read())
    block = re.search(r'extern\s+"C"\s*\{(.*?)\n\}', text, flags=re.S).group(1)
    decls = {}
    for m in re.finditer(r"fn\s+(phast_[a-z0-9_]+)\s*\((.*?)\)\s*(->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2), m.group(4)
        arglist = [a.strip() for a in args.split(",") if a.strip()]
        classes = [_rust_class(a.split(":", 1)[1]) for a in arglist]
        decls[name] = (_rust_class(ret) if ret else "void", classes)
    return decls


def test_extern_block_matches_the_c_header():
    protos, decls = c_prototypes(), rust_externs()
    assert len(decls) >= 20
    for name, (ret, args) in decls.items():
        assert name in protos, f"{name} is not declared in include/phastft_hip.h"
        c_ret, c_args = protos[name]
        assert len(args) == len(c_args), (name, args, c_args)
        assert ret == c_ret, (name, ret, c_ret)
        for i, (r, c) in enumerate(zip(args, c_args)):
            # a Rust `*const Opaque` may stand for any struct handle; scalar classes must agree exactly
            assert r == c, (name, i, r, c)


def test_every_bound_symbol_is_exported_by_the_library():
    from phastft_amd import build

    lib = C.CDLL(build.build())
    for name in rust_externs():
        assert hasattr(lib, name), name


def test_reference_module_paths_exist():
    src = os.path.join(CRATE, "src")
    read = lambda *p: open(os.path.join(src, *p)).read()
    lib_rs = read("lib.rs")
    # lib.rs:20-31 of the reference: algorithms private unless bench-internals; options and planner public
    assert re.search(r'#\[cfg\(feature = "bench-internals"\)\]\s*pub mod algorithms;', lib_rs)
    assert re.search(r'#\[cfg\(not\(feature = "bench-internals"\)\)\]\s*mod algorithms;', lib_rs)
    assert "pub mod options;" in lib_rs and "pub mod planner;" in lib_rs
    # lib.rs:23-27: complex_nums private with `complex-nums` alone, public with `bench-internals`
    assert re.search(r'#\[cfg\(feature = "bench-internals"\)\]\s*pub mod complex_nums;', lib_rs)
    assert re.search(r'#\[cfg\(all\(feature = "complex-nums", not\(feature = "bench-internals"\)\)\)\]\s*mod complex_nums;', lib_rs)
    cn = read("complex_nums.rs")
    for item in ("pub fn deinterleave<", "pub fn deinterleave_complex64(", "pub fn deinterleave_complex32(", "pub fn combine_re_im<"):
        assert item in cn, item
    # lib.rs:33-38: root re-exports
    for item in ("fft_32_dit_with_planner_and_opts", "fft_64_dit_with_planner_and_opts", "c2r_fft_f32",
                 "c2r_fft_f32_with_planner", "c2r_fft_f32_with_planner_and_scratch", "c2r_fft_f64",
                 "c2r_fft_f64_with_planner", "c2r_fft_f64_with_planner_and_scratch", "r2c_fft_f32",
                 "r2c_fft_f32_with_planner", "r2c_fft_f64", "r2c_fft_f64_with_planner"):
        assert re.search(r"pub use algorithms::(dit|r2c)::\{[^}]*\b" + item + r"\b", lib_rs, flags=re.S), item
    # lib.rs:143-226: the four planar entry points live at the crate root
    for item in ("fft_64_dit", "fft_32_dit", "fft_64_dit_with_planner", "fft_32_dit_with_planner"):
        assert re.search(r"impl_fft!\([^)]*\b" + item + r"\b", lib_rs, flags=re.S), item
    planner = read("planner.rs")
    for item in ("pub enum Direction", "pub enum PlannerMode", "PlannerDit64", "PlannerDit32", "PlannerR2c64",
                 "PlannerR2c32", "pub fn with_mode", "pub fn new"):
        assert item in planner, item
    # planner.rs:38-39: planners are Send + Sync values
    assert planner.count("unsafe impl Send for $name {}") == 2 and planner.count("unsafe impl Sync for $name {}") == 2
    assert "pub struct Options" in read("options.rs") and "pub fn guess_options" in read("options.rs")
    assert "pub mod bravo;" in read("algorithms", "mod.rs") and "pub mod dit;" in read("algorithms", "mod.rs")
    bravo = read("algorithms", "bravo.rs")
    assert "pub fn bit_rev_bravo_f64<S>(_simd: S, data: &mut [f64], n: usize)" in bravo
    assert "pub fn bit_rev_bravo_f32<S>(_simd: S, data: &mut [f32], n: usize)" in bravo
    cargo = open(os.path.join(CRATE, "Cargo.toml")).read()
    assert 'bench-internals = ["complex-nums"]' in cargo and 'complex-nums = ["dep:num-complex"]' in cargo
