#!/bin/bash
# A/B of an environment switch of the library inside ONE gpurun call, alternating (boxes differ by 3-5 %):
#   tools/ab_env.sh VAR "0 1" rounds -- command...
VAR=$1; VALS=$2; ROUNDS=$3; shift 4
for r in $(seq 1 $ROUNDS); do for v in $VALS; do echo "=== round $r $VAR=$v"; env $VAR=$v "$@" 2>&1 | grep -v amdgpu.ids; done; done
