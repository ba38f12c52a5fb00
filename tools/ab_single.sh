#!/bin/bash
# A/B of library variants on the HEADLINE (single f64 2^20, driver protocol K = 20 and K = 200), alternating, one call.
VARS=${1:-"_prevq ''"}; ROUNDS=${2:-3}
for r in $(seq 1 $ROUNDS); do
  for v in $VARS; do
    [ "$v" = "''" ] && v=""
    for k in 20 200; do
      PHASTFT_HIP_LIB=$PWD/phastft_amd/lib/libphastft_hip$v.so python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-configs --no-scaling-reference 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('round $r variant [$v] K=$k:', round(d['value'],2), 'GS/s', round(d['ms_per_step']*1e3,2), 'us', [round(x*1e3,2) for x in d['roofline']['pass_ms']])"
    done
  done
done
