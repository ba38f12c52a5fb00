#!/bin/bash
# Round 5, fifth GPU call: built-in wisdom for the 4096- / 8192-point twins (merged into the existing file), the 64-point real
# transforms with 16 points per thread (variant library, A/B + parity), the whole suite on the final library, the bench line.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
timeout 300 python tools/make_builtin_wisdom.py --lo 12 --hi 14 --budget-s 200 --merge phastft_amd/csrc/builtin_wisdom.inc --out $O/builtin_wisdom.inc --log $O/r05_wisdom_run_small.log > /dev/null 2>&1; echo "rc=$?" >> $O/r05_wisdom_run_small.log
if [ -s $O/builtin_wisdom.inc ]; then
    cp $O/builtin_wisdom.inc phastft_amd/csrc/builtin_wisdom.inc
    python -m phastft_amd.build > $O/rebuild.log 2>&1; echo "rc=$?" >> $O/rebuild.log
fi
V=$R/phastft_amd/lib/libphastft_hip_r5lp4.so
M=$R/phastft_amd/lib/libphastft_hip.so
for r in 1 2; do for lib in $M $V; do for tot in 22 27; do echo "=== round $r $(basename $lib) LADDER_TOTAL=$tot"; PHASTFT_HIP_LIB=$lib LADDER_TOTAL=$tot timeout 200 python tools/batch_ladder.py 6 7 2>&1 | grep -v amdgpu.ids; done; done; done > $O/r05_real64_lp4_ab.log
PHASTFT_HIP_LIB=$V timeout 600 python -m pytest tests -m gpu -q --timeout=600 -k "real or r2c or c2r" > $O/r05_real64_lp4_tests.log 2>&1; echo "rc=$?" >> $O/r05_real64_lp4_tests.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/full_tests.log 2>&1; echo "rc=$?" >> $O/full_tests.log
timeout 400 python bench.py > $O/r05_bench_default.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
grep -E "passed|failed" $O/full_tests.log $O/r05_real64_lp4_tests.log | tail -4; tail -2 $O/r05_wisdom_run_small.log; grep -E "^===|^2\^6" $O/r05_real64_lp4_ab.log
