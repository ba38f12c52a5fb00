#!/bin/bash
# Round 5, call 19: the whole -m gpu suite on the static rules alone (PHAST_BUILTIN_WISDOM=0), final tree.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
PHAST_BUILTIN_WISDOM=0 timeout 600 python -m pytest tests -m gpu -q --timeout=500 -p no:cacheprovider > $O/r05_gpu_tests_final_static_rules.log 2>&1; echo "rc=$?" >> $O/r05_gpu_tests_final_static_rules.log
grep -E "passed|failed|^FAILED" $O/r05_gpu_tests_final_static_rules.log | tail -5
