#!/bin/bash
# usage (GPU box): scripts/exp_preprocess_build_run.sh "<tag>" "<extra hipcc flags>"  -- rebuild with the flags, us per pre-processed image
tag=$1; flags=$2
python - <<PY
from epipolarconsistency_amd import build
build.build_library(force=True, extra_flags="$flags".split())
PY
echo "$tag: $(python scripts/bench_preprocess.py 2>/dev/null | head -1)"
