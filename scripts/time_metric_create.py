import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic
n, S, B = 400, 1024, 768
dev = torch.device("cuda", 0)
Ps = synthetic.short_scan(n, S, S, 0.308)
ctx = E.Context(0)
slabs = torch.rand((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
dtrs = [E.RadonIntermediate.wrap_device(ctx, slabs[k], B, B, S, S) for k in range(n)]
for r in range(3):
    ctx.synchronize(); t0 = time.perf_counter()
    m = E.MetricRadonIntermediate(ctx, Ps, dtrs)
    ctx.synchronize(); t1 = time.perf_counter()
    m.refreshRadonIntermediates(); ctx.synchronize(); t2 = time.perf_counter()
    v = m.evaluate(); t3 = time.perf_counter()
    print("create %.2f ms, refresh all %.2f ms, first evaluate %.2f ms" % (1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2)))
    m.close()
