#!/bin/bash
# Round 5, first GPU call: the new tests, the error budget, the whole -m gpu suite with every checked error recorded, a short
# run of the wisdom generator and the default bench line.  Every step has its own log under gpurun_out/ and its own timeout.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity_r5.py -q -s --timeout=600 > $O/r5_tests.log 2>&1; echo "rc=$?" >> $O/r5_tests.log
timeout 600 python tests/golden/make_error_budget.py $O/error_budget.json > $O/error_budget.log 2>&1; echo "rc=$?" >> $O/error_budget.log
PHAST_RECORD_ERRORS=$PWD/$O/recorded_errors.jsonl timeout 1200 python -m pytest tests -m gpu -q --ignore=tests/test_gpu_parity_r5.py --timeout=900 > $O/full_tests.log 2>&1; echo "rc=$?" >> $O/full_tests.log
timeout 400 python tools/make_builtin_wisdom.py --budget-s 200 --out $O/builtin_wisdom.inc --log $O/wisdom_run.log > /dev/null 2>&1; echo "rc=$?" >> $O/wisdom_run.log
timeout 400 python bench.py > $O/bench_default.log 2>&1; echo "rc=$?" >> $O/bench_default.log
tail -5 $O/r5_tests.log $O/full_tests.log $O/wisdom_run.log $O/bench_default.log
