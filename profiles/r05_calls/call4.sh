#!/bin/bash
# Round 5, fourth GPU call: the 4096-point twin for C2C (A/B after the capture_ready fix), the small end of the size ladder,
# the whole -m gpu suite on the final library, the default bench line.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
bash tools/ab_env.sh PHAST_SMALL_TWIN_MIN_LOG "13 12" 2 -- timeout 200 python tools/size_ladder.py 11 14 > $O/r05_small_twin_4096.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/full_tests.log 2>&1; echo "rc=$?" >> $O/full_tests.log
timeout 400 python bench.py > $O/r05_bench_default.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
grep -E "passed|failed" $O/full_tests.log | tail -2; tail -c 300 $O/r05_bench_default.json; grep -A4 "round 2" $O/r05_small_twin_4096.log | head -40
