#!/bin/bash
# Round 5, third GPU call (library with the generated built-in wisdom): the ladders with the wisdom on and off (same box), the
# 4096-point twin A/B, HBM-traffic counters of every bench configuration (FETCH_SIZE / WRITE_SIZE in separate passes,
# --kernel-trace only), the round-5 tests, the default bench line.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
(rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -v "^$") > $O/r05_box_clocks.txt
# 1. the regime between one transform and a full chip: 2^22 points in flight per call, wisdom on / off / on
for v in 1 0 1; do echo "=== PHAST_BUILTIN_WISDOM=$v"; PHAST_BUILTIN_WISDOM=$v LADDER_TOTAL=22 timeout 300 python tools/batch_ladder.py 6 22 2>&1 | grep -v amdgpu.ids; done > $O/r05_batch_ladder_2p22.log
for v in 1 0; do echo "=== PHAST_BUILTIN_WISDOM=$v"; PHAST_BUILTIN_WISDOM=$v timeout 400 python tools/size_ladder.py 10 26 2>&1 | grep -v amdgpu.ids; done > $O/r05_size_ladder.log
# 2. 4096 points on the multi-pass twin?
bash tools/ab_env.sh PHAST_SMALL_TWIN_MIN_LOG "13 12" 2 -- timeout 200 python tools/size_ladder.py 11 13 > $O/r05_small_twin_4096.log 2>&1
# 3. HBM traffic per launch of every configuration on the bench line
cp profiles/traffic_latest.json $O/traffic_latest.json
cd /tmp
pmc() {  # key, algorithmic bytes, out name, command...
    local key=$1 alg=$2 name=$3; shift 3
    rm -rf /tmp/prof_fetch /tmp/prof_write
    timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -- "$@" > /dev/null 2> /tmp/prof_fetch.err
    timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -- "$@" > /dev/null 2> /tmp/prof_write.err
    python $R/tools/summarize_prof.py pmc /tmp/prof_fetch /tmp/prof_write $O/r05_pmc_hbm_traffic_${name}.txt $O/traffic_latest.json $key "${*/$R\//}" $alg | tail -4
}
pmc single_2p20 33554432 single2p20 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-scaling-reference --no-configs
pmc single_2p26 2147483648 single2p26 python $R/tools/prof_workloads.py big --iters 4
pmc r2c_f32_2p24 134217728 r2c_f32_2p24 python $R/tools/prof_workloads.py r2c --iters 10
pmc c2r_f32_2p24 134217728 c2r_f32_2p24 python $R/tools/prof_workloads.py c2r --iters 10
pmc batch_2p20 34359738368 batch1024_2p20 python $R/tools/prof_workloads.py batch --batch 1024 --iters 3
pmc f32_2p20 16777216 f32_2p20 python $R/tools/prof_workloads.py single --dtype f32 --iters 20
pmc f32_2p26 1073741824 f32_2p26 python $R/tools/prof_workloads.py big --dtype f32 --iters 4
cd $R
cp $O/traffic_latest.json profiles/traffic_latest.json   # (bench.py below reads it)
# 4. the round-5 tests again (the fixed tune test), then the default line
timeout 900 python -m pytest tests/test_gpu_parity_r5.py -q -s --timeout=600 > $O/r5_tests.log 2>&1; echo "rc=$?" >> $O/r5_tests.log
timeout 400 python bench.py > $O/r05_bench_default.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
grep -E "passed|failed" $O/r5_tests.log | tail -2; tail -c 400 $O/r05_bench_default.json
