"""Where a small index-list evaluation spends its time on the host side (experiment)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic
n, S, B = 400, 1024, 768
dev = torch.device("cuda", 0)
Ps = synthetic.short_scan(n, S, S, 0.308)
torch.cuda.set_stream(torch.cuda.Stream(dev))  # a stream of our own, not the legacy default stream
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
dtrs = []
for a in range(0, n, 50):
    imgs = synthetic.projections_torch(Ps[a:a + 50], S, S, synthetic.sphere_phantom(), dev)
    dtrs += E.RadonIntermediate.compute_into(ctx, imgs, slabs[a:a + 50], B, B)
    ctx.synchronize()
P = E.pack_projection_matrices(Ps)
idx4 = np.array([(min(200, v), max(200, v), min(200, v), max(200, v)) for v in range(n) if v != 200], np.int32)
vals = np.empty(len(idx4), np.float32)
for small in (True, False):
    m = E.MetricRadonIntermediate(ctx, Ps, dtrs).setSampling("polynomial").setSmallEval(small)
    for what in ("set+eval(vals)", "eval(vals) only", "eval() only", "set only"):
        for _ in range(30):
            m.setProjectionMatrices(P); m.evaluate(idx4, vals)
        t0 = time.perf_counter()
        for _ in range(300):
            if what != "eval(vals) only" and what != "eval() only":
                m.setProjectionMatrices(P)
            if what == "set+eval(vals)" or what == "eval(vals) only":
                m.evaluate(idx4, vals)
            elif what == "eval() only":
                m.evaluate(idx4)
        print("small_eval %s: %-16s %.1f us" % (small, what, 1e6 * (time.perf_counter() - t0) / 300))
    m.close()
