#!/bin/bash
# Round 5, eleventh GPU call (experiment): the f32 wave tiles (build.py --experimental) as tuner candidates -- does the tuner
# find a call where a plan with 64 x 32 one-wave f32 tiles beats everything else by more than 3 %?
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
PHASTFT_HIP_LIB=$R/phastft_amd/lib/libphastft_hip_exp.so timeout 900 python tools/make_builtin_wisdom.py --dtypes f32 --kinds c2c,r2c,c2r --lo 14 --hi 25 --max-points 25 --budget-s 400 --out $O/wisdom_f32_wave.inc --log $O/r05_wisdom_f32_wave_tiles.log > /dev/null 2>&1
echo "rc=$?" >> $O/r05_wisdom_f32_wave_tiles.log
grep -c "w " $O/r05_wisdom_f32_wave_tiles.log; grep -E "p(8|16|32)w" $O/r05_wisdom_f32_wave_tiles.log | head -40; tail -2 $O/r05_wisdom_f32_wave_tiles.log
