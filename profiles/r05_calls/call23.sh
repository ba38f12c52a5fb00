#!/bin/bash
# Round 5, call 23: tests/cpp/tune_beside_callers_test.cpp plain, then the three threaded programs under ThreadSanitizer
# (tools/sanitize_host.sh run-tsan).
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
( echo "# $(date -u) plain: tests/cpp/tune_beside_callers_test"; timeout 40 tests/cpp/tune_beside_callers_test_plain; echo "# exit code $?"
  timeout 160 bash tools/sanitize_host.sh run-tsan ) > $O/r05_tsan_host.log 2>&1
grep -c "WARNING: ThreadSanitizer" $O/r05_tsan_host.log; cut -c1-700 $O/r05_tsan_host.log | head -60
