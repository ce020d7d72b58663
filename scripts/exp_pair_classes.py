"""Pair-kernel time per pair class on the BASELINE workload: index-list evaluations of (a) the pairs on the exact path,
(b) an equal number of degree-8 pairs, (c) degree-10 pairs.  python scripts/exp_pair_classes.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic
n, S, B = 400, 1024, 768
dev = torch.device("cuda", 0)
Ps = synthetic.short_scan(n, S, S, 0.308)
ph = synthetic.sphere_phantom()
ctx = E.Context(0)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
dtrs = []
for a in range(0, n, 50):
    imgs = synthetic.projections_torch(Ps[a:a + 50], S, S, ph, dev)
    dtrs += E.RadonIntermediate.compute_into(ctx, imgs, slabs[a:a + 50], B, B)
    ctx.synchronize()
m = E.MetricRadonIntermediate(ctx, Ps, dtrs).setSampling("polynomial")
deg = np.concatenate([[p["degree"] for p in m.debug_polynomials(a, min(10000, 79800 - a))] for a in range(0, 79800, 10000)])
iu = np.triu_indices(n, 1)
out = {}
rng = np.random.default_rng(0)
for name, sel in (("exact", deg == 0), ("deg8", deg == 8), ("deg10", deg == 10)):
    idx = np.flatnonzero(sel)
    idx = rng.choice(idx, size=28000, replace=True)
    idx4 = np.stack([iu[0][idx], iu[1][idx], iu[0][idx], iu[1][idx]], 1).astype(np.int32)
    vals = np.empty(len(idx4), np.float32)
    ctx.enable_timing(True)
    ks = []
    for _ in range(12):
        m.evaluate(idx4, vals)
        ks.append(ctx.last_kernel_ms("pairs"))
    ctx.enable_timing(False)
    out[name] = dict(pairs=len(idx4), kernel_ms=float(np.median(ks[2:])), us_per_1000_pairs=1e3 * float(np.median(ks[2:])) / len(idx4) * 1000)
print(json.dumps(out))
