#!/bin/bash
# usage (GPU box): scripts/exp_radon_build_run.sh "<tag>" "<extra hipcc flags>"  -- rebuild, bit-exactness check, ms per Radon intermediate
tag=$1; flags=$2
python - <<PY
from epipolarconsistency_amd import build
build.build_library(force=True, extra_flags="$flags".split())
PY
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "radon" 2>&1 | tail -1
echo "$tag: $(python scripts/bench_radon.py 50 1024 768 3 2>/dev/null | tail -1)"
