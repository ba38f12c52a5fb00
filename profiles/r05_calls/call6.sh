#!/bin/bash
# Round 5, sixth GPU call: the host code under ASan (incl. PlannerMode::Tune and the wisdom store, tests/cpp/host_api_test.cpp),
# the whole suite on the final library, the bench line plain and under rocprofv3 --stats.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
timeout 1500 bash tools/sanitize_host.sh run > $O/r05_asan_host_pass.log 2>&1; echo "# rc=$?" >> $O/r05_asan_host_pass.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/full_tests.log 2>&1; echo "rc=$?" >> $O/full_tests.log
timeout 400 python bench.py > $O/r05_bench_default.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
timeout 400 python bench.py --steps 20 --warmup 5 > $O/r05_bench_driver_protocol.json 2>> $O/bench.err
cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $R/bench.py --no-cpu-baseline > $O/r05_bench_default_under_rocprofv3.json 2> /tmp/prof_stats.err
cd $R && python tools/summarize_prof.py stats /tmp/prof_stats $O/r05_bench_default_kernel_stats.csv > /dev/null 2>&1
LADDER_TOTAL=27 timeout 400 python tools/batch_ladder.py 4 24 2>&1 | grep -v amdgpu.ids > $O/r05_batch_ladder.log
grep -E "passed|failed" $O/full_tests.log | tail -2; grep -E "exit code|ERROR|SUMMARY" $O/r05_asan_host_pass.log | head -20; tail -c 300 $O/r05_bench_default.json
