"""CPU-side checks of the drop-in boundary: libphastft_hip.so loads without a GPU, exports every symbol
include/phastft_hip.h declares, keeps the reference's panic strings, and never computes on the CPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "phastft_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(phast_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_listed():
    from phastft_amd import _lib

    lib = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 55
    for s in syms:
        getattr(lib, s)  # AttributeError = declared but not exported
    assert sorted(_lib.SYMBOLS) == syms  # the Python loader's list is the header's list


def test_header_cites_the_reference():
    text = open(HEADER).read()
    for cite in ("lib.rs:180", "lib.rs:143", "algorithms/dit.rs:263", "planner.rs:55", "planner.rs:194",
                 "algorithms/r2c.rs:521", "algorithms/r2c.rs:695", "options.rs:38", "algorithms/bravo.rs:303"):
        assert cite in text, cite


def test_strerror_keeps_reference_panic_text():
    from phastft_amd import _lib

    lib = _lib.lib()
    want = {4: "n must be a power of 2 >= 4", 5: "input length must match planner size",
            6: "output_re must have length N/2 + 1", 7: "output_im must have length N/2 + 1",
            8: "output length must match planner size", 9: "input_re must have length N/2 + 1",
            10: "input_im must have length N/2 + 1", 11: "scratch_re must have length N/2",
            12: "scratch_im must have length N/2"}
    for code, msg in want.items():
        assert lib.phast_strerror(code).decode() == msg


def test_host_logic_without_a_device():
    import phastft_amd as P

    o = P.Options.guess_options(1 << 20)  # options.rs:38-43
    assert o.multithreaded_bit_reversal and o.smallest_parallel_chunk_size == 16384
    assert not P.Options.guess_options(1 << 15).multithreaded_bit_reversal
    assert P.Options() == P.Options(False, 16384)  # options.rs:26-33
    assert int(P.Direction.Forward) == 1 and int(P.Direction.Reverse) == -1  # planner.rs:13-15
    with pytest.raises(P.PhastPanic):  # argument asserts fire before any device work
        P.PlannerDit64(5)
    with pytest.raises(P.PhastPanic) as ei:
        P.PlannerR2c32(6)
    assert str(ei.value) == "n must be a power of 2 >= 4"
    with pytest.raises(P.PhastPanic):
        P.bit_rev_bravo_f64(np.zeros(10), 3)
    with pytest.raises(TypeError):
        P.fft_64_dit(np.zeros(8, np.float32), np.zeros(8, np.float32), P.Direction.Forward)


def test_no_cpu_fallback():
    """Without a GPU every compute entry point must fail loudly, never compute on the host."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import phastft_amd as P

    re, im = np.arange(16.0), np.zeros(16)
    for call in (lambda: P.fft_64_dit(re, im, P.Direction.Forward), lambda: P.PlannerDit64(1024),
                 lambda: P.bit_rev_bravo_f64(np.arange(8.0), 3),
                 lambda: P.r2c_fft_f64(np.zeros(16), np.zeros(9), np.zeros(9))):
        with pytest.raises(P.PhastHipError):
            call()
    assert np.array_equal(re, np.arange(16.0))  # untouched


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "phastft_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("oracle/", "").replace("the oracle", "").replace("host oracle", "") \
                    or "import oracle" not in text, f
                assert "from oracle" not in text and "import oracle" not in text, f


def test_every_environment_variable_the_library_reads_is_documented():
    """INTEGRATION.md lists the environment variables; a getenv("PHAST...") in the sources that is not in that table is a
    behaviour switch a maintainer cannot find."""
    import glob
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for path in glob.glob(os.path.join(root, "phastft_amd", "csrc", "*")) + glob.glob(os.path.join(root, "phastft_amd", "*.py")):
        if os.path.isfile(path) and path.endswith((".hip", ".hpp", ".py")):
            names |= set(re.findall(r'getenv\("(PHAST[A-Z0-9_]*)"\)', open(path).read()))
            names |= set(re.findall(r'environ(?:\.get)?[\[(]\s*["\'](PHAST[A-Z0-9_]*)', open(path).read()))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    assert names, "no environment variables found: the pattern no longer matches the sources"
    missing = sorted(n for n in names if n not in doc)
    assert not missing, missing


def test_wisdom_store_without_a_device():
    """csrc/wisdom.hpp through the C ABI: import / export / forget need no GPU.  Later layers win, malformed lines are skipped,
    a text without the header is refused, and what comes out parses back to the same set."""
    import phastft_amd as P

    P.wisdom_forget()
    was = P.wisdom_builtin(True)
    assert P.wisdom_count(0) > 100         # the built-in layer (csrc/builtin_wisdom.inc) is there without a device, too ...
    base = P.wisdom_export()
    assert base == "phastft-hip-wisdom 1 cus=0\n"   # ... and never exported: a text that carried it would pin this build's plans
    P.wisdom_builtin(False)
    assert P.wisdom_count(-1) == 0
    text = ("phastft-hip-wisdom 1 cus=256\n"
            "f64 c2c 20 0 6,8,6@10,12,10:p8w fuse=0 us=23.10 heur=24.02\n"
            "f32 r2c 24 0 8,9,6@12,13,11:p16 fuse=1 us=88.00 heur=95.00\n"
            "f64 c2r 21 3 heuristic fuse=0 us=40.00 heur=40.00\n"
            "f64 c2c 99 0 6,8,6@10,12,10:p8w\n"          # length out of range: skipped
            "f64 c2c 20 1 6,8,6@10,12:p8 fuse=0\n"        # rows and tiles disagree: skipped
            "garbage line\n")
    P.wisdom_import(text)
    out = P.wisdom_export()
    for line in text.splitlines()[1:4]:
        assert line in out, line
    assert "garbage" not in out and " 99 " not in out and "f64 c2c 20 1" not in out
    # a later import of the same key replaces it
    P.wisdom_import("phastft-hip-wisdom 1 cus=256\nf64 c2c 20 0 7,7,6@12,12,12:p8 fuse=0 us=22.00 heur=24.00\n")
    out2 = P.wisdom_export()
    assert "f64 c2c 20 0 7,7,6@12,12,12:p8" in out2 and "6,8,6@10,12,10:p8w" not in out2
    # round trip
    P.wisdom_forget()
    P.wisdom_import(out2)
    assert P.wisdom_export() == out2
    with pytest.raises(P.PhastPanic):
        P.wisdom_import("f64 c2c 20 0 heuristic\n")  # no header
    with pytest.raises(P.PhastPanic):                 # another format version half way down: all or nothing
        P.wisdom_import("phastft-hip-wisdom 1 cus=256\nf64 c2c 22 0 heuristic fuse=0 us=1.00 heur=1.00\nphastft-hip-wisdom 2 cus=256\n")
    assert "f64 c2c 22 0" not in P.wisdom_export()
    P.wisdom_forget()
    assert P.wisdom_export() == base
    assert P.wisdom_builtin(True) is False   # the switch reports what it was ...
    assert P.wisdom_count(0) > 100 and P.wisdom_count(-1) == P.wisdom_count(0)
    # ... an entry of a later layer shadows the built-in one of the same key, and forget() brings the built-in plan back
    # (with one map for all layers the built-in entry was lost until the layer was switched on again)
    n0 = P.wisdom_count(0)
    key = next(ln for ln in open(os.path.join(ROOT, "phastft_amd", "csrc", "builtin_wisdom.inc")).read().splitlines()
               if ln.startswith('"f'))[1:].split(" fuse=")[0].rsplit(" ", 1)[0]
    P.wisdom_import("phastft-hip-wisdom 1 cus=256\n" + key + " heuristic fuse=0 us=1.00 heur=1.00\n")
    assert P.wisdom_count(0) == n0 - 1 and P.wisdom_count(2) == 1
    P.wisdom_forget()
    assert P.wisdom_count(0) == n0 and P.wisdom_count(-1) == n0
    # the header's arch= / lib= travel with the entries and come out again; a text without them comes out as it went in
    P.wisdom_import("phastft-hip-wisdom 1 cus=256 arch=gfx950 lib=6\nf64 c2c 20 0 7,7,6@12,12,12:p8 fuse=0 us=22.00 heur=24.00\n")
    assert P.wisdom_export().startswith("phastft-hip-wisdom 1 cus=256 arch=gfx950 lib=6\n")
    P.wisdom_forget()
    P.wisdom_builtin(was)


def test_wisdom_file_is_loaded_at_first_use(tmp_path):
    """PHAST_WISDOM=<path>: read when the store is first used (no device needed); PHAST_BUILTIN_WISDOM=0 leaves the built-in
    layer out.  (The rewrite after a tuning run needs a GPU: tests/test_gpu_parity_r5.py.)"""
    import subprocess
    import sys

    path = tmp_path / "wisdom.txt"
    path.write_text("phastft-hip-wisdom 1 cus=256\nf32 c2r 24 1 7,9,7@12,13,13:p16 fuse=0 us=185.13 heur=202.99\n")
    code = "import sys; sys.path.insert(0, %r)\nimport phastft_amd as P\nprint(P.wisdom_export())" % ROOT
    env = dict(os.environ, PHAST_WISDOM=str(path), PHAST_BUILTIN_WISDOM="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln and not ln.startswith("phastft-hip-wisdom")]
    assert lines == ["f32 c2r 24 1 7,9,7@12,13,13:p16 fuse=0 us=185.13 heur=202.99"], r.stdout


def test_wisdom_import_survives_arbitrary_text():
    """A wisdom file is input from outside (PHAST_WISDOM, phast_wisdom_import): whatever the text, the import either refuses it
    (no header / wrong version) or keeps exactly the lines that parse -- and what it kept comes out as text that imports back
    to the same set.  Lines are built from the tokens of the format, mutated: digits where names belong, lists too long, values
    out of range, stray separators."""
    import phastft_amd as P
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st

    tok = st.one_of(st.sampled_from(["f64", "f32", "c2c", "c2ci", "r2c", "c2r", "heuristic", "fuse=1", "fuse=0", "us=1.5", "heur=2",
                                     "6,6@10,11:p8", "9,9@13,13:p32", "6,8,6@10,12,10:p8w", "6,6,6,6@1,1,1,1:p8", "99,6@10,10:p8",
                                     "6,6@10,10:p7", "6,6@10:p8", ",@:p", "6,,6@10,10:p16", "phastft-hip-wisdom", "1", "2", "cus=256",
                                     "cus=", "#", "4294967296", "-1", "20", "0", "41"]),
                    st.text(alphabet="0123456789,@:pw=. -#\tfcru", max_size=12))
    line = st.lists(tok, max_size=9).map(" ".join)
    body = st.lists(line, max_size=12).map("\n".join)
    text = st.tuples(st.sampled_from(["phastft-hip-wisdom 1 cus=256\n", "phastft-hip-wisdom 1\n", "phastft-hip-wisdom 2 cus=1\n", ""]), body).map("".join)

    was = P.wisdom_builtin(False)
    try:
        @settings(max_examples=300, deadline=None, suppress_health_check=list(HealthCheck))
        @given(text)
        def check(t):
            P.wisdom_forget()
            try:
                P.wisdom_import(t)
            except P.PhastPanic:
                assert P.wisdom_count(-1) == 0, t   # a refused text leaves nothing behind ...
                return
            out = P.wisdom_export()
            n = P.wisdom_count(-1)
            assert n == len([ln for ln in out.splitlines() if ln and not ln.startswith("phastft-hip-wisdom")]), (t, out)
            P.wisdom_forget()
            P.wisdom_import(out)                    # ... and an accepted one is kept in the form it is written in
            assert P.wisdom_export() == out and P.wisdom_count(-1) == n, (t, out)

        check()
    finally:
        P.wisdom_forget()
        P.wisdom_builtin(was)


def test_no_cxx_exception_crosses_the_c_abi():
    """Callers are C, Rust (unwinding across `extern "C"` is undefined there) and ctypes: every entry point of csrc/c_abi.hip is a
    function-try-block.  The debug hook throws inside the library: std::bad_alloc comes back as PHAST_ERR_ALLOC (13), any
    other exception as PHAST_ERR_HIP (14) with its text in phast_last_hip_error() -- and the source has no entry point left
    without the barrier."""
    import re

    from phastft_amd import _lib

    l = _lib.lib()
    l.phast_debug_throw.restype = C.c_int
    l.phast_last_hip_error.restype = C.c_char_p
    assert l.phast_debug_throw(0) == 0
    assert l.phast_debug_throw(1) == 13
    assert l.phast_debug_throw(2) == 14 and b"phast_debug_throw" in l.phast_last_hip_error()
    assert l.phast_debug_throw(3) == 14 and b"unknown C++ exception" in l.phast_last_hip_error()
    src = open(os.path.join(ROOT, "phastft_amd", "csrc", "c_abi.hip")).read()
    body = src[src.index('extern "C" {'):]
    defs = re.findall(r"^\s*(?:int|size_t|void) (phast_[\w#]+)\([^;{]*?\)\s*(try\s*)?\{", body, re.M)
    assert len(defs) >= 55, len(defs)
    assert [name for name, tr in defs if not tr] == [], "entry points without the exception barrier"
