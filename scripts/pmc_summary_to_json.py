#!/usr/bin/env python3
"""Collects the per-kernel averages of the rocprofv3 --pmc passes (scripts/pmc_pass.sh / pmc_radon.sh write one
`*.summary.txt` per pass under gpurun_out/) into ONE json that bench.py reads for its roofline objects:

    python scripts/pmc_summary_to_json.py <tag> profiles/pmc_current.json gpurun_out/pmc_<tag>_*.summary.txt ...

Values are averages per dispatch as rocprofv3 reports them; the gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE
counts 128-B requests as 64 B) are applied by the reader, not here."""
import json, re, sys, time

tag, out = sys.argv[1], sys.argv[2]
kernels = {}
for path in sys.argv[3:]:
    for line in open(path):
        m = re.match(r"(\S.*?) ((?:[A-Za-z0-9_]+=[-+0-9.e]+ \(n=\d+\)(?:, )?)+)\s*$", line)
        if not m:
            continue
        name = m.group(1)
        for c, v, n in re.findall(r"([A-Za-z0-9_]+)=([-+0-9.e]+) \(n=(\d+)\)", m.group(2)):
            kernels.setdefault(name, {})[c] = float(v)
            kernels[name].setdefault("_dispatches", {})[c] = int(n)
json.dump({"tag": tag, "made": time.strftime("%Y-%m-%d"), "workload": "bench.py --steps 5 --warmup 2 (400 views, 1024^2, 768^2 bins, 1 GPU); "
           "radon_kernel: 50-image launches", "method": "rocprofv3 --kernel-trace --pmc <counters>, one pass per file "
           "(scripts/profile_round.sh), averages per dispatch", "files": [p.split("/")[-1] for p in sys.argv[3:]],
           "kernels": kernels}, open(out, "w"), indent=1, sort_keys=True)
print("wrote", out, "kernels:", sorted(kernels))
