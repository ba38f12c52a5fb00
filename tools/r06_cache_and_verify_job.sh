#!/bin/bash
# one gpurun call: (1) per-pass cache policy A/B of the 2^20 plan, (2) the built-in table re-tuned up to 2^27 points in flight,
# (3) every line of it replayed as a HIP graph against the static rule on the same buffers (tools/verify_builtin_wisdom.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out
timeout 900 python tools/ab_single.py --libs "'',_As0,_Cl0,_Al0,_scr,_ntl0q,_nts0q" --cases f64:20,f32:20 --rounds 3 > $O/r06_cache_policy_ab.log 2>&1
timeout 1500 python tools/make_builtin_wisdom.py --max-points 27 --budget-s 1200 --out $O/builtin_wisdom.inc --log $O/r06_wisdom_run.log > $O/wisdom_run.stdout 2>&1
tail -2 $O/r06_wisdom_run.log
timeout 2400 python tools/verify_builtin_wisdom.py --inc $O/builtin_wisdom.inc --imported --out $O/builtin_wisdom.verified.inc --log $O/r06_wisdom_verify.log > $O/wisdom_verify.stdout 2>&1
tail -5 $O/r06_wisdom_verify.log
