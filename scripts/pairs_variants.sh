#!/bin/bash
# usage (GPU box): scripts/pairs_variants.sh "<hipcc flags>" ...  -- rebuilds per flag set, prints the pair-kernel time of bench.py
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  python3 - <<PY
from epipolarconsistency_amd import build
build.build_library(force=True, extra_flags="$v".split())
PY
  echo "== $v"
  python3 bench.py --steps 300 --warmup 100 --no-cpu-baseline 2>gpurun_out/pv_err.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel_ms', d['roofline']['kernel_ms'], 'evals/s', d['value'], 'value', d['last_value'])"
done
python3 -m epipolarconsistency_amd.build --force > /dev/null
