"""BASELINE config 5 on one GPU: FDCTMotionCorrection-style inner loop.  View n/2 of the 400-view scan is
perturbed by the "3D Rigid" parameters (ref: LibProjectiveGeometry/Models/ModelSimilarity3D.hxx:64-88,
P' = P T) and swept like plotCostFunction does (ref: Gui/Visualization.h:78-98): 6 parameters x 100
steps over [-5, 5] mm / [-2, 2] deg = 600 full all-pairs evaluations.  Reports evaluations/s and checks
6 sampled sweep points against the CPU oracle."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import geometry, synthetic

n, S, B = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (400, 1024, 768)
dev = torch.device("cuda", 0)
Ps = synthetic.short_scan(n, S, S, 0.308 * 1024 / S)
ph = synthetic.sphere_phantom()
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
dtrs = []
for a in range(0, n, 50):
    imgs = synthetic.projections_torch(Ps[a:a + 50], S, S, ph, dev)
    dtrs += E.RadonIntermediate.compute_into(ctx, imgs, slabs[a:a + 50], B, B)
    ctx.synchronize()
m = E.MetricRadonIntermediate(ctx, Ps, dtrs)
moving = n // 2
names = ["tx", "ty", "tz", "rx", "ry", "rz"]
ranges = [5.0, 5.0, 5.0] + [np.deg2rad(2.0)] * 3
packed = E.pack_projection_matrices(Ps)
P0 = Ps[moving].copy()
m.evaluate()  # warm-up
values = np.zeros((6, 100))
t0 = time.perf_counter()
for p in range(6):
    for k in range(100):
        x = -ranges[p] + 2 * ranges[p] * k / 99.0
        packed[moving] = (P0 @ geometry.rigid_transform(**{names[p]: x})).T.reshape(12)
        m.setProjectionMatrices(packed)
        values[p, k] = m.evaluate()
elapsed = time.perf_counter() - t0
out = {"config": "config 5: %d views %dx%d, view %d swept over 6 rigid parameters x 100 steps" % (n, S, S, moving),
       "evaluations": 600, "seconds": elapsed, "evaluations_per_s": 600 / elapsed, "n_gpus": 1,
       "min_at_step": [int(np.argmin(values[p])) for p in range(6)]}
# parity at 6 sampled sweep points
import oracle
oracle.build(native=True)
host = [d.readback() for d in dtrs]
errs = []
for p, k in [(0, 7), (1, 49), (2, 93), (3, 20), (4, 50), (5, 81)]:
    x = -ranges[p] + 2 * ranges[p] * k / 99.0
    Pk = list(Ps)
    Pk[moving] = P0 @ geometry.rigid_transform(**{names[p]: x})
    ref = oracle.evaluate_all(Pk, host, S, S, native=True)["mean"]
    errs.append(abs(values[p, k] - ref) / abs(ref))
out["parity_rel_err_at_6_points"] = errs
print(json.dumps(out))
