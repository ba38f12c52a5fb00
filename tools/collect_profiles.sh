#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel statistics, HBM-traffic counters (FETCH_SIZE / WRITE_SIZE in separate
# passes) and SQ counters of the bench workloads, summarised into gpurun_out/profiles_new/ (copy what should be
# judged into profiles/).  Counter passes use --kernel-trace only (never the hip/hsa trace domains).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profiles_new
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r03}
# clocks / power state of the box beside every collection: boxes differ by 3-5 % (copy probe 4.75-5.23 TB/s), so numbers
# from different collections compare only through these and the same-run copy probe
(rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -v "^$"; echo; cat /sys/class/drm/card*/device/pp_dpm_mclk 2>/dev/null) > $O/${TAG}_box_clocks.txt
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM"
# 1. the driver's command under the profiler: per-kernel statistics of everything on the default line
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $R/bench.py --no-cpu-baseline > $O/${TAG}_bench_default_under_rocprofv3.json 2> /tmp/prof_stats.err
python $R/tools/summarize_prof.py stats /tmp/prof_stats $O/${TAG}_bench_default_kernel_stats.csv > /dev/null
# 2. HBM traffic: headline (single 2^20), N = 2^26, R2C f32 2^24
pmc() {  # key, algorithmic bytes, out name, command...
    local key=$1 alg=$2 name=$3; shift 3
    rm -rf /tmp/prof_fetch /tmp/prof_write
    timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -- "$@" > /dev/null 2> /tmp/prof_fetch.err
    timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -- "$@" > /dev/null 2> /tmp/prof_write.err
    python $R/tools/summarize_prof.py pmc /tmp/prof_fetch /tmp/prof_write $O/${TAG}_pmc_hbm_traffic_${name}.txt $O/traffic_latest.json $key "${*/$R\//}" $alg | tail -4
}
pmc single_2p20 33554432 single2p20 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-scaling-reference --no-configs
pmc single_2p26 2147483648 single2p26 python $R/tools/prof_workloads.py big --iters 4
pmc r2c_f32_2p24 134217728 r2c_f32_2p24 python $R/tools/prof_workloads.py r2c --iters 10
pmc c2r_f32_2p24 134217728 c2r_f32_2p24 python $R/tools/prof_workloads.py c2r --iters 10
pmc batch_2p20 34359738368 batch1024_2p20 python $R/tools/prof_workloads.py batch --batch 1024 --iters 3
pmc f32_2p20 16777216 f32_2p20 python $R/tools/prof_workloads.py single --dtype f32 --iters 20
pmc f32_2p26 1073741824 f32_2p26 python $R/tools/prof_workloads.py big --dtype f32 --iters 4
# 3. where the wave cycles go (SQ counters, one pass of 8)
for w in single big; do
    rm -rf /tmp/prof_sq
    timeout 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d /tmp/prof_sq -- python $R/tools/prof_workloads.py $w --iters 5 > /dev/null 2> /tmp/prof_sq.err
    python $R/tools/summarize_sq.py /tmp/prof_sq $O/${TAG}_sq_${w}.txt "python tools/prof_workloads.py $w --iters 5"
done
# 4. kernel statistics of the other workloads
for w in batch bitrev; do
    rm -rf /tmp/prof_wl
    # the batch is profiled in the shape the bench runs it: 1024-transform launches (round 2 profiled 256-transform
    # launches with a fill kernel between them and read 4.05 TB/s where the bench's launches gave 5.0)
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wl -- python $R/tools/prof_workloads.py $w --batch 1024 --iters 6 > $O/${TAG}_${w}.log 2> /tmp/prof_wl.err
    python $R/tools/summarize_prof.py stats /tmp/prof_wl $O/${TAG}_${w}_kernel_stats.csv > /dev/null
done
# 5. the plain (un-profiled) default line, for comparison with the profiled one
cd $R && timeout 600 python bench.py > $O/${TAG}_bench_default_plain.json 2> $O/bench_plain.err
head -c 1500 $O/${TAG}_bench_default_kernel_stats.csv
