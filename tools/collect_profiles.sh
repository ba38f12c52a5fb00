#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel statistics and HBM-traffic counters of the headline bench,
# summarised into gpurun_out/profiles_new/ (copy what should be judged into profiles/).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profiles_new
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r01}
CMD="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-scaling-reference"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $R/bench.py --no-cpu-baseline --no-scaling-reference > $O/${TAG}_bench_single2p20_under_rocprofv3.json 2> /tmp/prof_stats.err
python $R/tools/summarize_prof.py stats /tmp/prof_stats $O/${TAG}_bench_single2p20_kernel_stats.csv > /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -- $CMD > /dev/null 2> /tmp/prof_fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -- $CMD > /dev/null 2> /tmp/prof_write.err
python $R/tools/summarize_prof.py pmc /tmp/prof_fetch /tmp/prof_write $O/${TAG}_pmc_hbm_traffic_single2p20.txt $O/traffic_latest.json single_2p20 \
    "python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-scaling-reference" | tail -5
# the other workloads (batch shard, 2^26, R2C, bit reversal): kernel statistics only
for w in batch big r2c bitrev; do
    rm -rf /tmp/prof_wl
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wl -- python $R/tools/prof_workloads.py $w > $O/${TAG}_${w}.log 2> /tmp/prof_wl.err
    python $R/tools/summarize_prof.py stats /tmp/prof_wl $O/${TAG}_${w}_kernel_stats.csv > /dev/null
done
cd $R && timeout 300 python bench.py --extra > $O/${TAG}_bench_single2p20_plain_extra.json 2> $O/bench_extra.err
head -c 600 $O/${TAG}_bench_single2p20_kernel_stats.csv
