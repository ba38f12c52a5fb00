#!/bin/bash
# Round 5, second GPU call: (1) the wisdom generator over every bucket up to 2^25 points in flight -> builtin_wisdom.inc, the
# library rebuilt with it ON THE BOX; (2) the whole -m gpu suite with that library (round-5 tests included), every checked
# error recorded; (3) the default bench line; (4) rocprofv3 kernel statistics of the bench command.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 700 python tools/make_builtin_wisdom.py --budget-s 420 --out $O/builtin_wisdom.inc --log $O/wisdom_run.log > /dev/null 2>&1; echo "rc=$?" >> $O/wisdom_run.log
if [ -s $O/builtin_wisdom.inc ]; then
    cp $O/builtin_wisdom.inc phastft_amd/csrc/builtin_wisdom.inc
    python -m phastft_amd.build > $O/rebuild.log 2>&1; echo "rc=$?" >> $O/rebuild.log
fi
PHAST_RECORD_ERRORS=$PWD/$O/recorded_errors.jsonl timeout 1500 python -m pytest tests -m gpu -q -s --timeout=900 > $O/full_tests.log 2>&1; echo "rc=$?" >> $O/full_tests.log
timeout 400 python bench.py > $O/bench_default.log 2>&1; echo "rc=$?" >> $O/bench_default.log
cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprofv3.json 2> /tmp/prof_stats.err
cd $GRAFT_REPO_ROOT && python tools/summarize_prof.py stats /tmp/prof_stats $O/r05_bench_default_kernel_stats.csv > /dev/null 2>&1
grep -E "passed|failed" $O/full_tests.log | tail -3; tail -2 $O/wisdom_run.log; tail -c 600 $O/bench_default.log
