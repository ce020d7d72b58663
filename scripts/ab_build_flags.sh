#!/bin/bash
# usage (GPU box): scripts/ab_build_flags.sh "<python command line>" "<hipcc flags A>" "<hipcc flags B>" ...
# Rebuilds the library with each flag set in turn (twice round: A B ... A B ...) and runs the command after each build;
# the last build is the default one again.
cd $GRAFT_REPO_ROOT
cmd=$1; shift
for round in 1 2; do
  for v in "$@"; do
    python3 - <<PY
from epipolarconsistency_amd import build
build.build_library(force=True, extra_flags="$v".split())
PY
    echo "== flags: '$v'"
    bash -c "$cmd" 2>/dev/null
  done
done
python3 -m epipolarconsistency_amd.build --force > /dev/null
