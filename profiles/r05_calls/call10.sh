#!/bin/bash
# Round 5, tenth GPU call: SQ counters of the headline's and the 2^26 transform's kernels, kernel statistics of the 1024-transform
# batch and of bit reversal (tools/collect_profiles.sh steps 3 and 4), on the final library.
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM"
for w in single big; do
    rm -rf /tmp/prof_sq
    timeout 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d /tmp/prof_sq -- python $R/tools/prof_workloads.py $w --iters 5 > /dev/null 2> /tmp/prof_sq.err
    python $R/tools/summarize_sq.py /tmp/prof_sq $O/r05_sq_${w}.txt "python tools/prof_workloads.py $w --iters 5"
done
for w in batch bitrev; do
    rm -rf /tmp/prof_wl
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wl -- python $R/tools/prof_workloads.py $w --batch 1024 --iters 6 > $O/r05_${w}.log 2> /tmp/prof_wl.err
    python $R/tools/summarize_prof.py stats /tmp/prof_wl $O/r05_${w}_kernel_stats.csv > /dev/null
done
cd $R && timeout 300 python -m pytest tests/test_gpu_parity_r5.py -q -x --timeout=600 > $O/r5_tests.log 2>&1; echo "rc=$?" >> $O/r5_tests.log
head -12 $O/r05_sq_single.txt; head -4 $O/r05_batch_kernel_stats.csv | cut -c1-160; tail -2 $O/r5_tests.log
