#!/bin/bash
# Round 5's library (lib/libphastft_hip_base.so = HEAD b929417, its own built-in wisdom) against this round's on ONE box, alternating:
# the size ladder (one transform per call, every entry point) and the full-chip batch ladder -- is any call slower?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out
for lib in _base "" _base ""; do
    echo "=== library [${lib:-product}] (size_ladder 12 26)"; PHASTFT_HIP_LIB=$PWD/phastft_amd/lib/libphastft_hip$lib.so timeout 400 python tools/size_ladder.py 12 26 2>&1 | grep -v amdgpu.ids
done > $O/r06_vs_r05_size_ladder.log
for r in 1 2; do for lib in _base ""; do
    echo "=== round $r library [${lib:-product}] (LADDER_TOTAL=27, batch_ladder 13 22)"; PHASTFT_HIP_LIB=$PWD/phastft_amd/lib/libphastft_hip$lib.so LADDER_TOTAL=27 timeout 400 python tools/batch_ladder.py 13 22 2>&1 | grep -v amdgpu.ids
done; done > $O/r06_vs_r05_batch_ladder.log
tail -20 $O/r06_vs_r05_size_ladder.log | cut -c1-170
