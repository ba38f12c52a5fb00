#!/bin/bash
# Round 5, call 16: a graph captured before a tuning run keeps replaying the plan it captured.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
for i in 1 2; do
timeout 300 python -m pytest tests/test_gpu_parity_r5.py -m gpu -q -s --timeout=250 -p no:cacheprovider -k "graph_captured_before" 2>&1 | grep -v amdgpu.ids
done > $O/r05_graph_across_tune.log 2>&1
tail -40 $O/r05_graph_across_tune.log | cut -c1-800
