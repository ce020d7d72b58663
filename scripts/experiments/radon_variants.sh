#!/bin/bash
# usage (GPU box): scripts/radon_variants.sh "-DRT_TILE_W=64 -DRT_TILE_H=64" ...  -- rebuilds per flag set, times the Radon kernel
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  python3 - <<PY
from epipolarconsistency_amd import build
build.build_library(force=True, extra_flags="$v".split())
PY
  echo "== $v"
  python3 scripts/bench_radon.py 50 1024 768 3 2>&1 | tail -1
done
python3 -m epipolarconsistency_amd.build --force > /dev/null
