#!/bin/bash
# Round 5, call 18 (the last): the whole -m gpu suite, smoke(), the default bench line and the driver's protocol on the library
# as committed; then the round's profile set (rocprofv3 kernel statistics of the bench command, PMC traffic, SQ counters).
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q --timeout=500 -p no:cacheprovider > $O/r05_gpu_tests_final.log 2>&1; echo "rc=$?" >> $O/r05_gpu_tests_final.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke.log 2>&1; echo "rc=$?" >> $O/r05_smoke.log
timeout 300 python bench.py > $O/r05_bench_default.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 > $O/r05_bench_driver_protocol.json 2>> $O/bench.err
grep -E "passed|failed" $O/r05_gpu_tests_final.log | tail -2; tail -2 $O/r05_smoke.log; tail -c 300 $O/r05_bench_default.json
timeout 560 bash tools/collect_profiles.sh r05 > $O/collect.log 2>&1; echo "collect rc=$?"
ls $O/profiles_new | head -40
