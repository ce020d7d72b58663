"""Time of ONE launch over exactly the pairs on the per-sample path of the BASELINE workload (the kappa_max = pi/2 pairs, in
their natural order), with the row-paired copies and with the row-quad copies (Context.debugSetQuadCopies): what a second,
heavy-only launch after a main launch that skips them would cost.  python scripts/exp_heavy_alone.py"""
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
out = {}
for quads in (0, 1):
    if quads:
        ctx.debugSetQuadCopies(True)
    m = E.MetricRadonIntermediate(ctx, Ps, dtrs).setSampling("polynomial")
    deg = np.concatenate([[p["degree"] for p in m.debug_polynomials(a, min(10000, 79800 - a))] for a in range(0, 79800, 10000)])
    iu = np.triu_indices(n, 1)
    idx = np.flatnonzero(deg == 0)
    idx4 = np.stack([iu[0][idx], iu[1][idx], iu[0][idx], iu[1][idx]], 1).astype(np.int32)
    vals = np.empty(len(idx4), np.float32)
    ctx.enable_timing(True)
    ks = []
    for _ in range(14):
        m.evaluate(idx4, vals)
        ks.append(ctx.last_kernel_ms("pairs"))
    ctx.enable_timing(False)
    out["quads" if quads else "paired"] = dict(pairs=len(idx4), kernel_us=1e3 * float(np.median(ks[2:])))
    m.close()
print(json.dumps(out))
