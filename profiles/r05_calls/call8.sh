#!/bin/bash
# Round 5, eighth GPU call: the built-in wisdom regenerated with the final tuner (result check, aligned ring, buckets up to 2^26
# points in flight, lengths from 4096), the library rebuilt with it on the box, the whole suite with the wisdom on AND off, the
# default bench line, SQ counters of the headline.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
timeout 900 python tools/make_builtin_wisdom.py --max-points 26 --budget-s 600 --out $O/builtin_wisdom.inc --log $O/r05_wisdom_run.log > /dev/null 2>&1; echo "rc=$?" >> $O/r05_wisdom_run.log
cp $O/wisdom_full.txt $O/r05_wisdom_full.txt 2>/dev/null
if [ -s $O/builtin_wisdom.inc ] && grep -q "^# .* tuning runs" $O/r05_wisdom_run.log; then
    cp $O/builtin_wisdom.inc phastft_amd/csrc/builtin_wisdom.inc
    python -m phastft_amd.build > $O/rebuild.log 2>&1; echo "rc=$?" >> $O/rebuild.log
fi
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/full_tests.log 2>&1; echo "rc=$?" >> $O/full_tests.log
PHAST_BUILTIN_WISDOM=0 timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --deselect tests/test_abi.py > $O/full_tests_static_rules.log 2>&1; echo "rc=$?" >> $O/full_tests_static_rules.log
timeout 400 python bench.py > $O/r05_bench_default.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
grep -E "passed|failed" $O/full_tests.log $O/full_tests_static_rules.log | tail -4; tail -2 $O/r05_wisdom_run.log; grep -c "FAILED the result" $O/r05_wisdom_run.log; tail -c 300 $O/r05_bench_default.json
