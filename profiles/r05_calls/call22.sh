#!/bin/bash
# Round 5, call 22: the host code (c_abi.hip built -fsanitize=thread, host side only) under ThreadSanitizer -- four threads x four
# streams on one planner, then the stress program (eight threads, three planners, streams destroyed in between), 50 s each.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
export TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 second_deadlock_stack=1 suppressions=$R/tools/tsan.supp exitcode=0"
( echo "# $(date -u) TSan pass 1: tests/cpp/concurrent_planner_test_tsan"; timeout 50 tests/cpp/concurrent_planner_test_tsan; echo "# exit code $?"
  echo "# $(date -u) TSan pass 2: tests/cpp/planner_stress_test_tsan"; timeout 50 tests/cpp/planner_stress_test_tsan; echo "# exit code $?" ) > $O/r05_tsan_host.log 2>&1
grep -c "WARNING: ThreadSanitizer" $O/r05_tsan_host.log; grep -E "^# |WARNING: ThreadSanitizer|SUMMARY" $O/r05_tsan_host.log | sort | uniq -c | sort -rn | head -20
grep -A14 -m1 "WARNING: ThreadSanitizer" $O/r05_tsan_host.log | cut -c1-200
