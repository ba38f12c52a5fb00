#!/bin/bash
# Round 5, call 14: the new GPU test alone -- every built-in wisdom line at its own call, against the static rule's plan and numpy.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity_r5.py -m gpu -q -s --timeout=500 -p no:cacheprovider -k "every_builtin_wisdom or wisdom_text or torch_free" > $O/r05_builtin_wisdom_parity.log 2>&1; echo "rc=$?" >> $O/r05_builtin_wisdom_parity.log
tail -25 $O/r05_builtin_wisdom_parity.log | cut -c1-400
