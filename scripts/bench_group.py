"""Single-process multi-GPU (ecc_group_*): evaluations/s of the BASELINE workload through ONE process, one host thread
per device -- the form a C++ caller of the adapter gets.  Usage (GPU box):
    python scripts/bench_group.py <devices, e.g. 0 or 0,1,2,3 or 0,0> [views size bins steps]
Repeated device ids put several ranks (stream + thread each) on one GPU: a rehearsal of the hand-off cost, not a
scaling number."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import geometry, synthetic

devices = [int(d) for d in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
n, S, B, steps = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (400, 1024, 768, 300)
dev = torch.device("cuda", devices[0])
torch.cuda.set_device(dev)
Ps = synthetic.short_scan(n, S, S, 0.308 * 1024 / S)
ph = synthetic.sphere_phantom()
ctx = E.Context(devices[0], stream=torch.cuda.current_stream().cuda_stream)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
dtrs = []
for a in range(0, n, 50):
    imgs = synthetic.projections_torch(Ps[a:a + 50], S, S, ph, dev)
    dtrs += E.RadonIntermediate.compute_into(ctx, imgs, slabs[a:a + 50], B, B)
    ctx.synchronize()
g = E.Group(devices)
t0 = time.perf_counter()
gm = E.GroupMetricRadonIntermediate(g, Ps, dtrs)
setup_s = time.perf_counter() - t0
single = E.MetricRadonIntermediate(ctx, Ps, dtrs)
packed = E.pack_projection_matrices(Ps)
moving, P0 = n // 2, Ps[n // 2].copy()
poses = []
for k in range(64):
    Pk = packed.copy()
    Pk[moving] = (P0 @ geometry.rigid_transform(tx=0.01 * k, rz=1e-4 * (k % 7))).T.reshape(12)
    poses.append(Pk)
ref = single.setProjectionMatrices(poses[5]).evaluate()
got = gm.setProjectionMatrices(poses[5]).evaluate()
out = {"devices": devices, "views": n, "size": S, "bins": B, "replication_and_setup_s": setup_s,
       "rel_diff_vs_single_context": abs(got - ref) / ref}
for name, m in (("group", gm), ("single_context", single)):
    for k in range(30):
        m.setProjectionMatrices(poses[k % 64]).evaluate()
    blocks = []
    for b in range(5):
        t0 = time.perf_counter()
        for k in range(steps):
            m.setProjectionMatrices(poses[k % 64]).evaluate()
        blocks.append((time.perf_counter() - t0) / steps)
    out[name + "_ms_per_step"] = 1e3 * sorted(blocks)[2]
    out[name + "_evaluations_per_s"] = 1.0 / sorted(blocks)[2]
# BASELINE config 5 through the group: 600 independent sweep poses, pose p on rank p mod G (no exchange)
sweep = [poses[k % 64] for k in range(600)]
gm.evaluate_poses(sweep[:16])
t0 = time.perf_counter()
vals = gm.evaluate_poses(sweep)
el = time.perf_counter() - t0
out["config5_poses_per_s"] = 600 / el
out["config5_check"] = bool(vals[5] == single.setProjectionMatrices(poses[5]).evaluate())
print(json.dumps(out))
