"""Diagnostics of the -DPK_EXP_LDS_STAGE build (GPU box): fraction of 64-sample trips that took the LDS-staged path per
degree class, from the counters the experimental kernel leaves in the K01 debug output; plus parity of the mean."""
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
n_pairs = n * (n - 1) // 2
K = m.debug_K01(0, n_pairs)
deg = np.array([d["degree"] for d in m.debug_polynomials(0, 20000)])
lds, all_ = K[:, 0].astype(np.float64), K[:, 1].astype(np.float64)
out = {"mean": m.evaluate(), "trips_total": float(all_.sum()), "trips_on_lds_path": float(lds.sum()),
       "fraction": float(lds.sum() / max(all_.sum(), 1))}
for d in (4, 6, 8, 10):
    sel = np.where(deg == d)[0]
    if len(sel):
        out["fraction_degree_%d" % d] = float(lds[sel].sum() / max(all_[sel].sum(), 1))
print(json.dumps(out))
