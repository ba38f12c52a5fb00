#!/bin/bash
# Round 5, call 20: the library with the exception barrier at the C ABI -- the whole -m gpu suite, smoke() and the default bench line.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
timeout 420 python -m pytest tests -m gpu -q --timeout=400 -p no:cacheprovider > $O/r05_gpu_tests_final.log 2>&1; echo "rc=$?" >> $O/r05_gpu_tests_final.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke.log 2>&1; echo "rc=$?" >> $O/r05_smoke.log
timeout 200 python bench.py > $O/r05_bench_default.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
grep -E "passed|failed|^FAILED" $O/r05_gpu_tests_final.log | tail -3; tail -2 $O/r05_smoke.log; head -c 400 $O/r05_bench_default.json
