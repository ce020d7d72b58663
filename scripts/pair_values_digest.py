"""SHA-256 of all pair values of a 400-view scan (random Radon intermediates, 768 x 768 bins) in the polynomial and the
per-sample mode: a refactoring that must not change results is checked by running this before and after (GPU box)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic

n, S, B = 400, 1024, 768
dev = torch.device("cuda", 0)
ctx = E.Context(0)
Ps = synthetic.short_scan(n, S, S, 0.308)
g = torch.Generator(device=dev).manual_seed(7)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
imgs = torch.rand((8, 256, 256), generator=g, device=dev)
small = torch.zeros((8, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
E.RadonIntermediate.compute_into(ctx, imgs, small, B, B)  # eight real dtrs (borders replicated as the layout wants), cycled
ctx.synchronize()
for v in range(n):
    slabs[v] = small[v % 8] * (1.0 + 0.01 * v)
dtrs = [E.RadonIntermediate.wrap_device(ctx, slabs[v], B, B, S, S) for v in range(n)]
m = E.MetricRadonIntermediate(ctx, Ps, dtrs)
for mode in ("polynomial", "per_sample"):
    m.setSampling(mode)
    total, vals = m.evaluate_range(0, n * (n - 1) // 2, want_pairs=True)
    print(mode, hashlib.sha256(vals.tobytes()).hexdigest(), repr(total))
