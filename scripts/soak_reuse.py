"""Soak of the record-reuse path of the headline problem (GPU box): 400 views at 768 x 768 bins, one view moved per step over a
cycle of poses, every mean compared with the one a metric without reuse and without the small-evaluation paths gave for the
same pose.  usage: soak_reuse.py [steps, default 20000] [views, default 400]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic, geometry

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
S, B = 1024, 768
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(dev))
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
Ps = synthetic.short_scan(n, S, S, 0.308)
g = torch.Generator(device=dev).manual_seed(3)
small = torch.zeros((8, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
E.RadonIntermediate.compute_into(ctx, torch.rand((8, 256, 256), generator=g, device=dev), small, B, B)
ctx.synchronize()
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
for v in range(n):
    slabs[v] = small[v % 8] * (1.0 + 0.01 * v)
dtrs = [E.RadonIntermediate.wrap_device(ctx, slabs[v], B, B, S, S) for v in range(n)]
P = E.pack_projection_matrices(Ps)
poses = []
for k in range(48):
    Pk = P.copy()
    v = (n // 2, n // 3, 5)[k % 3]  # the moved view changes too: records of two views to refit at once now and then
    Pk[v] = (Ps[v] @ geometry.rigid_transform(tx=0.01 * (k % 16), rz=1e-4 * (k % 7))).T.reshape(12)
    poses.append(Pk)
ref = E.MetricRadonIntermediate(ctx, Ps, dtrs).setRecordReuse(False).setSmallEval(False)
want = [ref.setProjectionMatrices(Pk).evaluate() for Pk in poses]
ref.close()
m = E.MetricRadonIntermediate(ctx, Ps, dtrs)
bad, t0 = 0, time.time()
for it in range(steps):
    k = it % 48
    v = m.setProjectionMatrices(poses[k]).evaluate()
    if v != want[k]:
        bad += 1
        if bad < 10:
            print("MISMATCH step %d pose %d: %r != %r" % (it, k, v, want[k]), flush=True)
    if (it + 1) % 5000 == 0:
        print("%d steps, %d mismatches, %.1f s" % (it + 1, bad, time.time() - t0), flush=True)
print("soak: %d steps of %d pairs, %d mismatches, %.3f ms per step" % (steps, n * (n - 1) // 2, bad, 1e3 * (time.time() - t0) / steps))
sys.exit(1 if bad else 0)
