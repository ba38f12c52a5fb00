#!/bin/bash
# Round 5, call 24: the FINAL host code under AddressSanitizer -- pass 1 (the C++ host test incl. tuner and wisdom store, planner
# cache off, leak check on) and pass 6 (a tuning run beside three calling threads) of tools/sanitize_host.sh.  The *_asan lines of
# .gpurunignore were lifted for this call only.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
export LSAN_OPTIONS=suppressions=$R/tools/lsan.supp:print_suppressions=0
( echo "# $(date -u) host-side ASan pass 1 (PHAST_PLANNER_CACHE=0, leak check on): tests/cpp/host_api_test_asan gpu"
  ASAN_OPTIONS=detect_leaks=1:halt_on_error=0:abort_on_error=0 PHAST_PLANNER_CACHE=0 timeout 60 tests/cpp/host_api_test_asan gpu; echo "# exit code $?"
  echo "# $(date -u) host-side ASan pass 6 (a tuning run and a Tune-mode planner beside three calling threads; leak check on)"
  ASAN_OPTIONS=detect_leaks=1:halt_on_error=0:abort_on_error=0 timeout 60 tests/cpp/tune_beside_callers_test_asan; echo "# exit code $?" ) > $O/r05_asan_final.log 2>&1
grep -E "^# |ERROR|SUMMARY|passed|failed|\{" $O/r05_asan_final.log | cut -c1-500 | tail -20
