#!/bin/bash
# Round 5, call 13: the whole -m gpu suite twice more on the committed library -- a flake hunt before the driver's own run.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
for i in 1 2; do
  timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $O/flake_run$i.log 2>&1; echo "rc=$?" >> $O/flake_run$i.log
done
grep -E "passed|failed|^FAILED" $O/flake_run1.log $O/flake_run2.log
