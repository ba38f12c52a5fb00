set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -f $O/r06_errors.jsonl
PHAST_WISDOM_AB_LOG=$PWD/$O/r06_wisdom_vs_static_ab.log PHAST_RECORD_ERRORS=$PWD/$O/r06_errors.jsonl timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/r06_gpu_tests_full.log
timeout 900 python tests/golden/make_error_budget.py $O/error_budget.json > $O/r06_error_budget.log 2>&1
timeout 1500 bash tools/collect_profiles.sh r06 > $O/r06_collect.log 2>&1
cat $O/r06_gpu_tests_full.log; tail -3 $O/r06_error_budget.log; tail -5 $O/r06_collect.log
