# The round's closing GPU job: the whole -m gpu suite (recording the error budget and the wisdom A/B), the refreshed error budget,
# then every profile of profiles/README.md (tools/collect_profiles.sh) and the driver's own command.
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -f $O/r06_errors.jsonl
PHAST_WISDOM_AB_LOG=$PWD/$O/r06_wisdom_vs_static_ab.log PHAST_RECORD_ERRORS=$PWD/$O/r06_errors.jsonl timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/r06_gpu_tests_full.log
cat $O/r06_gpu_tests_full.log | tail -4
# (tools/r06_vs_r05_ladders.sh ran here until the Complex<T> <-> planes entry points were added: the Python package binds every
#  symbol of the header at load, which round 5's library does not have -- the ladders of record are profiles/r06_vs_r05_*_ladder.log)
timeout 900 python tests/golden/make_error_budget.py $O/error_budget.json > $O/r06_error_budget.log 2>&1
timeout 1500 bash tools/collect_profiles.sh r06 > $O/r06_collect.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > $O/r06_bench_driver_protocol.json 2> $O/bench_driver.err
tail -3 $O/r06_error_budget.log; tail -5 $O/r06_collect.log
