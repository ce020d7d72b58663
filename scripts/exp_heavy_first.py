"""Experiment: does the ORDER in which the kappa_max = pi/2 pairs are dispatched matter?  All 79 800 pairs of the BASELINE
workload as an index list in (a) natural get_ij order, (b) heavy pairs first, (c) heavy pairs last, (d) heavy pairs spread
evenly.  Kernel time by HIP events.  QUADS=1 in the environment of this script to combine with the row-quad copies (Context.debugSetQuadCopies)."""
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
if os.environ.get("QUADS") == "1":
    ctx.debugSetQuadCopies(True)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
dtrs = []
for a in range(0, n, 50):
    imgs = synthetic.projections_torch(Ps[a:a + 50], S, S, ph, dev)
    dtrs += E.RadonIntermediate.compute_into(ctx, imgs, slabs[a:a + 50], B, B)
    ctx.synchronize()
m = E.MetricRadonIntermediate(ctx, Ps, dtrs).setSampling("polynomial")
N = n * (n - 1) // 2
km = np.concatenate([m.debug_K01(a, min(10000, N - a))[:, 15] for a in range(0, N, 10000)])
heavy = km > np.pi / 4
iu = np.triu_indices(n, 1)
nat = np.arange(N)
orders = {"natural": nat, "heavy_first": np.concatenate([nat[heavy], nat[~heavy]]), "heavy_last": np.concatenate([nat[~heavy], nat[heavy]])}
# spread: one heavy pair every N/H positions
H = int(heavy.sum())
spread = np.empty(N, np.int64)
pos_h = (np.arange(H) * (N / H)).astype(np.int64)
mask = np.zeros(N, bool); mask[pos_h] = True
spread[mask] = nat[heavy]; spread[~mask] = nat[~heavy]
orders["heavy_spread"] = spread
# heavy pairs as whole workgroups at the start: the mapping gives wave w of block b pair w*nblk + b, so put heavy pairs at
# positions {w*nblk + b : b < H/4}
nblk = (N + 3) // 4
hb = (H + 3) // 4
posw = np.concatenate([w * nblk + np.arange(hb) for w in range(4)])[:H]
mask = np.zeros(N, bool); mask[posw] = True
wg = np.empty(N, np.int64); wg[mask] = nat[heavy]; wg[~mask] = nat[~heavy]
orders["heavy_first_whole_workgroups"] = wg
out = {"heavy_pairs": H, "quad_copies": os.environ.get("QUADS", "0")}
vals = np.empty(N, np.float32)
ref = None
for name, o in orders.items():
    idx4 = np.stack([iu[0][o], iu[1][o], iu[0][o], iu[1][o]], 1).astype(np.int32)
    ctx.enable_timing(True)
    ks = []
    for _ in range(15):
        mean = m.evaluate(idx4, vals)
        ks.append(ctx.last_kernel_ms("pairs"))
    ctx.enable_timing(False)
    out[name] = float(np.median(ks[3:]))
    ref = mean if ref is None else ref
    assert abs(mean - ref) <= 1e-12 * ref
print(json.dumps(out))
