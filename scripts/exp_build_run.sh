#!/bin/bash
# usage (GPU box): scripts/exp_build_run.sh "<tag>" "<extra hipcc flags>" [env assignments...]   -- rebuilds the library with the
# flags and runs scripts/exp_pairs.py; one JSON line per call
tag=$1; flags=$2; shift 2
python - <<PY
from epipolarconsistency_amd import build
build.build_library(force=True, extra_flags="$flags".split())
PY
env "$@" python scripts/exp_pairs.py "$tag" 2>/dev/null | grep "^{"
