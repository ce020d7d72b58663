"""ecc_metric_evaluate_poses (two deep) against one pose at a time on the BASELINE data set (400 views, 1024^2): poses/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic, geometry
n, S, B = 400, 1024, 768
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(dev))
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
Ps = synthetic.short_scan(n, S, S, 0.308)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
dtrs = []
for a in range(0, n, 50):
    imgs = synthetic.projections_torch(Ps[a:a + 50], S, S, synthetic.sphere_phantom(), dev)
    dtrs += E.RadonIntermediate.compute_into(ctx, imgs, slabs[a:a + 50], B, B)
    ctx.synchronize()
m = E.MetricRadonIntermediate(ctx, Ps, dtrs)
P = E.pack_projection_matrices(Ps)
poses = []
for k in range(200):
    Pk = P.copy()
    Pk[200] = (Ps[200] @ geometry.rigid_transform(tx=0.01 * (k % 50), rz=1e-4 * (k % 7))).T.reshape(12)
    poses.append(Pk)
for rep in range(3):
    t0 = time.perf_counter(); a = m.evaluate_poses(poses); t1 = time.perf_counter()
    b = np.array([m.setProjectionMatrices(Pk).evaluate() for Pk in poses]); t2 = time.perf_counter()
    print("two deep %.1f poses/s (%.1f us per pose), one at a time %.1f poses/s (%.1f us); identical: %s"
          % (200 / (t1 - t0), 5e3 * (t1 - t0), 200 / (t2 - t1), 5e3 * (t2 - t1), np.array_equal(a, b)), flush=True)
