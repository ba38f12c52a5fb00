#!/bin/bash
# Round 5, call 25: the pytest wrapper of tests/cpp/tune_beside_callers_test.cpp (twice).
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export TMPDIR=/tmp
for i in 1 2; do timeout 120 python -m pytest tests/test_gpu_parity_r5.py -m gpu -q -s --timeout=100 -p no:cacheprovider -k "cpp_tuning_beside" 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-600; done
