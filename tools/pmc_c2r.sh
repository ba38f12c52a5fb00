# PMC traffic + kernel statistics of the C2R workload, fused and unfused (runs on the GPU box; profiles/README.md)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/profiles_new; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
pmc() {
    local key=$1 alg=$2 name=$3; shift 3
    rm -rf /tmp/prof_fetch /tmp/prof_write
    timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -- "$@" > /dev/null 2> /tmp/prof_fetch.err
    timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -- "$@" > /dev/null 2> /tmp/prof_write.err
    python $R/tools/summarize_prof.py pmc /tmp/prof_fetch /tmp/prof_write $O/r03_pmc_hbm_traffic_${name}.txt $O/traffic_tmp.json $key "${*/$R\//}" $alg | tail -6
}
pmc c2r_f32_2p24 134217728 c2r_f32_2p24 python $R/tools/prof_workloads.py c2r --iters 10
PHAST_C2R_FUSE=0 pmc c2r_f32_2p24_unfused 134217728 c2r_f32_2p24_unfused python $R/tools/prof_workloads.py c2r --iters 10
rm -rf /tmp/prof_wl; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wl -- python $R/tools/prof_workloads.py c2r --iters 20 > /dev/null 2>/tmp/e.err
python $R/tools/summarize_prof.py stats /tmp/prof_wl $O/r03_c2r_f32_2p24_kernel_stats.csv | head -8
