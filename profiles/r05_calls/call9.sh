#!/bin/bash
# Round 5, ninth GPU call: the built-in wisdom from the final tuner (the static rule's own plan is no candidate, more rounds for
# short calls), the library rebuilt with it on the box, the whole suite, what a planner costs to make, the default bench line.
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
timeout 300 python tools/planner_cost.py 2>&1 | grep -v amdgpu.ids > $O/r05_planner_cost.log
timeout 400 python bench.py > $O/r05_bench_default.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke.log 2>&1; echo "rc=$?" >> $O/r05_smoke.log
grep -E "passed|failed" $O/full_tests.log | tail -2; tail -2 $O/r05_wisdom_run.log; cat $O/r05_planner_cost.log; tail -3 $O/r05_smoke.log; tail -c 200 $O/r05_bench_default.json
