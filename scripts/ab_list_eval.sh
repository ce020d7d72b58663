#!/bin/bash
# usage (GPU box): scripts/ab_list_eval.sh  -- the moved view's list launch as one kernel (ecc_launch_list_eval) against k01_kernel +
# pairs_split_kernel: step time at shard size and at N = 1, pose-delta sweep.  A/B/A on one box (rebuilds the library twice).
cd $GRAFT_REPO_ROOT
run() {
  python3 scripts/step_fixed_cost.py 8 300 2>/dev/null | grep -A1 "reuse True"
  python3 scripts/step_fixed_cost.py 1 200 2>/dev/null | grep -A1 "reuse True"
  ECC_SWEEP_INCREMENTAL=1 python3 scripts/config5_sweep.py 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pose-delta sweep', d['incremental']['evaluations_per_s'], 'full', d['evaluations_per_s'])"
}
for v in "-DECC_LIST_EVAL_MAX_PAIRS=0" "" "-DECC_LIST_EVAL_MAX_PAIRS=0" ""; do
  python3 - <<PY
from epipolarconsistency_amd import build
build.build_library(force=True, extra_flags="$v".split())
PY
  echo "== flags: '$v'"
  run
done
