#!/bin/bash
# Round 5, call 21: the rebuilt library (comment-only change in a header) -- smoke(), the round-5 GPU tests and the bench line.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 150 python -m pytest tests/test_gpu_parity_r5.py tests/test_gpu_fuzz.py -m gpu -q --timeout=120 -p no:cacheprovider -x 2>&1 | tail -2
timeout 60 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | head -c 200
