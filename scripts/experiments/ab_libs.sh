#!/bin/bash
# usage (GPU box): scripts/experiments/ab_libs.sh <tag> [<tag> ...]  -- bench.py once per tag, in the order given (repeat a tag for
# A/B/A/B), each with ECC_HIP_LIB=scripts/experiments/_build/libecc_<tag>.so; one summary line per run.  BENCH=<script> runs that
# instead of bench.py (scripts/experiments/bench_with_quads.py)
i=0
for t in "$@"; do i=$((i+1)); ECC_HIP_LIB=$PWD/scripts/experiments/_build/libecc_$t.so python ${BENCH:-bench.py} --no-cpu-baseline --no-live-pmc --no-power > gpurun_out/ab_${i}_$t.json 2>gpurun_out/ab_${i}_$t.err; done
python - "$@" <<PY
import json, sys
for i, t in enumerate(sys.argv[1:], 1):
    try:
        d = json.loads(open("gpurun_out/ab_%d_%s.json" % (i, t)).read().strip().splitlines()[-1])
        print(t, round(d["value"], 1), "step", round(d["ms_per_step"], 4), "kernel", round(d["roofline"].get("kernel_ms"), 4), "nonpair", round(d["non_pair_kernel_us_per_step"], 1), d.get("last_value"))
    except Exception as e:
        print(t, "failed", e)
PY
