"""SHA-256 of all pair values of a 400-view scan (random Radon intermediates, 768 x 768 bins) in the polynomial and the
per-sample mode: a refactoring that must not change results is checked by running this before and after (GPU box).
With the argument `reference`: also the reference arithmetic on the first 6000 pairs (one wave per pair), the first 2000
(four waves per pair) and three lists of 1 / 300 / 512 pairs (the small and wide forms)."""
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
if "reference" in sys.argv[1:]:
    import numpy as np
    m.setSampling("reference")
    for first, count in ((0, 6000), (0, 2000), (40000, 300), (123, 512), (79799, 1)):
        total, vals = m.evaluate_range(first, count, want_pairs=True)
        print("reference", first, count, hashlib.sha256(vals.tobytes()).hexdigest(), repr(total))
    idx = [(min(200, v), max(200, v)) for v in range(n) if v != 200]
    idx4 = np.array([(a, b, a, b) for a, b in idx], np.int32)
    vals = np.empty(len(idx4), np.float32)
    print("reference list 399", repr(m.evaluate(idx4, vals)), hashlib.sha256(vals.tobytes()).hexdigest())
