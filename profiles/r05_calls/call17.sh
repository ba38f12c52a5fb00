#!/bin/bash
# Round 5, call 17: does a HOT ring (3 sets = 96 MiB f64 / 48 MiB f32: resident in the 256 MiB Infinity Cache, as the bench's in-place
# loop is) rank the plans of one 2^20-point transform differently from the cold ring the tuner uses?
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
for ring in 1 1280; do
for rep in 1 2; do
PHAST_BUILTIN_WISDOM=0 PHAST_TUNE_RING_MB=$ring PHAST_TUNE_MIN_GAIN_PERMILLE=10 PHAST_TUNE_ROUNDS=7 PHAST_TUNE_FINALS=6 timeout 200 python - <<PY 2>&1 | grep -v amdgpu.ids
import phastft_amd as P
for name, Pl in (("f64", P.PlannerDit64), ("f32", P.PlannerDit32)):
    for L in (20, 21):
        pl = Pl(1 << L)
        r = pl.tune(1)
        print("ring_mb=$ring", name, L, r["adopted"], r["plan"], "heur %.2f best %.2f us" % (r["us_heuristic"], r["us_best"]), r["candidates"], "plans %.2f s" % r["seconds"])
        P.wisdom_forget()
PY
done
done > $O/r05_hot_ring_tune.log 2>&1
cat $O/r05_hot_ring_tune.log | cut -c1-300
