#!/bin/bash
# The seeded fuzz of tests/test_gpu_fuzz.py over OTHER seeds, four times the cases per chunk (each C2C case against float64 pocketfft on every
# bin, the padding between transforms untouched; R2C / C2R batches against rfft / irfft): python -m pytest per seed, one line each.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for seed in ${@:-1 2 3 4 5 6 7 8}; do
    echo -n "PHAST_FUZZ_SEED=$seed PHAST_FUZZ_SCALE=4: "
    PHAST_FUZZ_SEED=$seed PHAST_FUZZ_SCALE=4 timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -1
done
