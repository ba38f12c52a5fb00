#!/bin/bash
# copy models of ONE transform of 2^20 points (tools/pass_probe.hip): what the passes of the plans would cost with free
# arithmetic, and with a pass's arithmetic as dependent FMAs + LDS round trips.  Run on the GPU box.
cd ${GRAFT_REPO_ROOT:-.}
P=tools/pass_probe.bin
echo "# f32, the plan that runs: 64x64 / 256x16 / 64x64 at 8 points per thread"
$P f32 6 8 6 64 16 64 8 0 0 4
$P f32 6 8 6 64 16 64 8 10 2 4
echo "# f32, middle pass with 128-byte rows (256 x 32 on 1024 threads)"
$P f32 6 8 6 64 32 64 8 0 0 1
echo "# f32, 128 x 64 x 128: every pass with >= 128-byte rows"
$P f32 7 6 7 32 64 32 8 0 0 4
$P f32 7 6 7 32 64 32 8 10 2 4
echo "# f64 for reference: 64x64 / 256x16 / 64x64 at 8 points per thread (the latency plan)"
$P f64 6 8 6 64 16 64 8 0 0 2
$P f64 6 8 6 64 16 64 8 10 2 2
