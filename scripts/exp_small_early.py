"""Where the latency of a one-launch evaluation goes (experiment): ECC_SMALL_DBG_MODE makes workgroup 0 report at kernel
entry (1), after the records (2), after its value (3) instead of the last arriver after the sum (0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic
n, S, B = 400, 1024, 768
dev = torch.device("cuda", 0)
Ps = synthetic.short_scan(n, S, S, 0.308)
torch.cuda.set_stream(torch.cuda.Stream(dev))
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
dtrs = []
for a in range(0, n, 50):
    imgs = synthetic.projections_torch(Ps[a:a + 50], S, S, synthetic.sphere_phantom(), dev)
    dtrs += E.RadonIntermediate.compute_into(ctx, imgs, slabs[a:a + 50], B, B)
    ctx.synchronize()
P = E.pack_projection_matrices(Ps)
rng = np.random.default_rng(0)
lists = {"1 pair": [(10, 250)], "399 pairs of view 200": [(min(200, v), max(200, v)) for v in range(n) if v != 200],
         "399 random pairs": [tuple(sorted(rng.choice(n, 2, replace=False))) for _ in range(399)]}
m = E.MetricRadonIntermediate(ctx, Ps, dtrs).setSampling("polynomial")
for name, idx in lists.items():
    idx4 = np.array([(a, b, a, b) for a, b in idx], np.int32)
    row = []
    for mode in (1, 2, 3, 0):
        os.environ["ECC_SMALL_DBG_MODE"] = str(mode)
        for _ in range(30):
            m.setProjectionMatrices(P); m.evaluate(idx4); ctx.synchronize()
        t = 0.0
        for _ in range(200):
            m.setProjectionMatrices(P)
            t0 = time.perf_counter()
            m.evaluate(idx4)
            t += time.perf_counter() - t0
            ctx.synchronize()  # the early report leaves the kernel running
        row.append(1e6 * t / 200)
    print("%-24s entry %.1f us, records %.1f, value %.1f, sum (normal) %.1f" % (name, *row))
