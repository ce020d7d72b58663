"""Step time (setProjectionMatrices + evaluate, record reuse on: the moved views' pairs are refitted and sampled by list launches on the
side stream beside the all-pairs launch) by the number of views that move per step.  python scripts/exp_moved_views.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic
n, S, B = 400, 1024, 768
dev = torch.device("cuda", 0)
Ps = synthetic.short_scan(n, S, S, 0.308)
ph = synthetic.sphere_phantom()
torch.cuda.set_stream(torch.cuda.Stream(dev))
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
for a in range(0, n, 50):
    imgs = synthetic.projections_torch(Ps[a:a + 50], S, S, ph, dev)
    keep = E.RadonIntermediate.compute_into(ctx, imgs, slabs[a:a + 50], B, B)
    ctx.synchronize()
dtrs = [E.RadonIntermediate.wrap_device(ctx, slabs[k], B, B, S, S) for k in range(n)]
m = E.MetricRadonIntermediate(ctx, Ps, dtrs)
P = E.pack_projection_matrices(Ps)
out = {}
for moved in (0, 1, 2, 4, 8, 16, 32, 64):
    P2 = P.copy()
    views = np.linspace(20, 380, max(moved, 1)).astype(int)[:moved]
    for v in views:
        P2.reshape(-1)[v * 12 + 9] += 1e-3
    for k in range(30):
        m.setProjectionMatrices(P2 if k & 1 else P); m.evaluate()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    steps = 600
    for k in range(steps):
        m.setProjectionMatrices(P2 if k & 1 else P); val = m.evaluate()
    torch.cuda.synchronize()
    out[moved] = round(1e6 * (time.perf_counter() - t0) / steps, 1)
print(json.dumps(out))
