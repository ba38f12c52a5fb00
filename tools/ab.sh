#!/bin/bash
# A/B of library variants inside ONE gpurun call (boxes differ by 3-5 %: numbers from different calls do not compare).
#   tools/ab.sh "<variant suffixes, '' = product>" "<f64 sizes>" "<f32 sizes>" [rounds]
# alternates the variants `rounds` times; each run is tools/cmp_throughput.py (best of 3 x 3 timed repetitions).
VARS=${1:-"_prev ''"}; F64=${2:-"20x1024 26x1 24x4"}; F32=${3:-""}; ROUNDS=${4:-2}
for r in $(seq 1 $ROUNDS); do
  for v in $VARS; do
    [ "$v" = "''" ] && v=""
    echo "=== round $r variant [$v]"
    [ -n "$F64" ] && PHASTFT_HIP_LIB=$PWD/phastft_amd/lib/libphastft_hip$v.so python tools/cmp_throughput.py $F64 2>&1 | grep "^2\^"
    [ -n "$F32" ] && PHASTFT_HIP_LIB=$PWD/phastft_amd/lib/libphastft_hip$v.so python tools/cmp_throughput.py --f32 $F32 2>&1 | grep "^2\^" | sed 's/^/f32 /'
  done
done
