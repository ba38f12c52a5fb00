#!/bin/bash
# Round 5, call 15: a tuning run beside three threads that transform with the same planner (three times: a flake hunt).
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
for i in 1 2 3; do
timeout 300 python -m pytest tests/test_gpu_parity_r5.py -m gpu -q -s --timeout=250 -p no:cacheprovider -k "tuning_while_other_threads" 2>&1 | grep -v amdgpu.ids
done > $O/r05_tune_beside_callers.log 2>&1
tail -30 $O/r05_tune_beside_callers.log | cut -c1-600
