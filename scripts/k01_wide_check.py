"""k01_kernel<8> (launches of at most ECC_K01_WIDE_MAX_PAIRS pairs) against k01_kernel<1> on the BASELINE geometry: the fitted
records and the pair values of sub-ranges must be bit-identical to those of the full launch; timing of a 399-pair index
list and of a 9 975-pair shard (GPU box)."""
import json, os, sys, time
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
total, vals = m.evaluate_range(0, n_pairs, want_pairs=True)       # one launch of 79 800 pairs: k01_kernel<1>
out = {"pairs": n_pairs}
bad = 0
for first, count in ((0, 5000), (20000, 9975), (70000, 9800), (399, 399), (12345, 7)):
    t, v = m.evaluate_range(first, count, want_pairs=True)          # k01_kernel<8>
    bad += int(not np.array_equal(v, vals[first:first + count]))
out["sub_ranges_differing"] = bad
P = E.pack_projection_matrices(Ps)
for name, first, count in (("shard_9975", 29925, 9975), ("range_399", 1000, 399)):
    for _ in range(20):
        m.setProjectionMatrices(P); m.evaluate_range(first, count)
    t0 = time.perf_counter()
    for _ in range(300):
        m.setProjectionMatrices(P); m.evaluate_range(first, count)
    out[name + "_us_per_step"] = 1e6 * (time.perf_counter() - t0) / 300
print(json.dumps(out))
