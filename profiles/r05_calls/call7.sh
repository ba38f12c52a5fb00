#!/bin/bash
# Round 5, seventh GPU call: (1) round 4's library against this round's on ONE box -- the full-chip batch ladder and the size
# ladder, alternating: does the refactor (table cache, choose, workspace rules) or the wisdom cost anything where no plan
# changed?  (2) the host code under ASan.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
for r in 1 2; do
  for tree in _r04 .; do
    echo "=== round $r tree $tree (LADDER_TOTAL=27, batch_ladder 13 22)"; (cd $R/$tree && LADDER_TOTAL=27 timeout 300 python tools/batch_ladder.py 13 22 2>&1 | grep -v amdgpu.ids)
  done
done > $O/r05_vs_r04_batch_ladder.log
for tree in _r04 . _r04 .; do
    echo "=== tree $tree (size_ladder 12 24)"; (cd $R/$tree && timeout 300 python tools/size_ladder.py 12 24 2>&1 | grep -v amdgpu.ids)
done > $O/r05_vs_r04_size_ladder.log
timeout 1500 bash tools/sanitize_host.sh run > $O/r05_asan_host_pass.log 2>&1; echo "# rc=$?" >> $O/r05_asan_host_pass.log
grep -E "exit code|ERROR|SUMMARY|failure" $O/r05_asan_host_pass.log | head -20
grep -E "^===|^2\^(14|16|20) " $O/r05_vs_r04_batch_ladder.log | cut -c1-150
