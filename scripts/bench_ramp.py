"""Ramp-filtered Radon intermediates (Filter::Ramp, 8f-3): ms per 1024^2 -> 768^2 dtr, split into the line-integral kernel
and the ramp filter, from a rocprofv3-free wall clock (stream synchronised) against the derivative filter."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic
n, S, B = 50, 1024, 768
dev = torch.device("cuda", 0)
Ps = synthetic.short_scan(400, S, S, 0.308)[:n]
imgs = synthetic.projections_torch(Ps, S, S, synthetic.sphere_phantom(), dev)
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
for name, f in (("derivative", E.FILTER_DERIVATIVE), ("none", E.FILTER_NONE), ("ramp", E.FILTER_RAMP)):
    for r in range(3):
        ctx.synchronize()
        t0 = time.perf_counter()
        keep = E.RadonIntermediate.compute_into(ctx, imgs, slabs, B, B, filter=f)
        ctx.synchronize()
        dt = time.perf_counter() - t0
    print("%-10s %.3f ms per Radon intermediate (wall clock, %d images)" % (name, 1e3 * dt / n, n))
