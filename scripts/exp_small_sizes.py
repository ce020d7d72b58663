"""One-launch path against the stream-ordered launches by evaluation size (all pairs of n views, one view moved per step):
us per setProjectionMatrices + evaluate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic, geometry
S, B = 512, 768
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(dev))
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
N = 91
Ps_all = synthetic.short_scan(N, S, S, 0.616)
slabs = torch.zeros((N, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
imgs = synthetic.projections_torch(Ps_all, S, S, synthetic.sphere_phantom(), dev)
dtrs_all = E.RadonIntermediate.compute_into(ctx, imgs, slabs, B, B)
ctx.synchronize()
for n in [int(a) for a in sys.argv[1:]] or [2, 8, 20, 33, 46, 64, 91]:
    Ps = Ps_all[:n]
    P = E.pack_projection_matrices(Ps)
    poses = []
    for k in range(64):
        Pk = P.copy()
        Pk[n // 2] = (Ps[n // 2] @ geometry.rigid_transform(tx=0.01 * k)).T.reshape(12)
        poses.append(Pk)
    row = []
    for small in (True, False):
        m = E.MetricRadonIntermediate(ctx, Ps, dtrs_all[:n]).setSmallEval(small)
        for k in range(50):
            m.setProjectionMatrices(poses[k % 64]); m.evaluate()
        t0 = time.perf_counter()
        for k in range(400):
            m.setProjectionMatrices(poses[k % 64]); m.evaluate()
        row.append(1e6 * (time.perf_counter() - t0) / 400)
        m.close()
    print("n = %3d (%4d pairs): one launch %.1f us, stream-ordered launches %.1f us" % (n, n * (n - 1) // 2, row[0], row[1]), flush=True)
