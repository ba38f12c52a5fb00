#!/bin/bash
# Round 5, last GPU call: the whole -m gpu suite, smoke() and the default bench line on the library as committed.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/full_tests.log 2>&1; echo "rc=$?" >> $O/full_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke.log 2>&1; echo "rc=$?" >> $O/r05_smoke.log
timeout 400 python bench.py > $O/r05_bench_default.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
timeout 400 python bench.py --steps 20 --warmup 5 > $O/r05_bench_driver_protocol.json 2>> $O/bench.err
grep -E "passed|failed" $O/full_tests.log | tail -2; tail -2 $O/r05_smoke.log; tail -c 200 $O/r05_bench_default.json
