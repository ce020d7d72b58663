"""A/B of the pair kernel's XCD schedule (ecc_debug_set_xcd_schedule) on one box: kernel time by HIP events of the all-pairs
launch and of the middle rank's cost-balanced shard of 2 / 4 / 8-rank jobs, and the wall-clock step (one moved view + evaluate),
with the table off / on (/ other segment lengths), interleaved.  python scripts/ab_xcd_schedule.py [modes, e.g. 0,1,0,1,16,400]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import sharding, synthetic
modes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1,0,1").split(",")]
worlds = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
n, S, B = 400, 1024, 768
dev = torch.device("cuda", 0)
Ps = synthetic.short_scan(n, S, S, 0.308)
ph = synthetic.sphere_phantom()
torch.cuda.set_stream(torch.cuda.Stream(dev))
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
for a in range(0, n, 50):
    imgs = synthetic.projections_torch(Ps[a:a + 50], S, S, ph, dev)
    keep = E.RadonIntermediate.compute_into(ctx, imgs, slabs[a:a + 50], B, B)
    ctx.synchronize()
dtrs = [E.RadonIntermediate.wrap_device(ctx, slabs[k], B, B, S, S) for k in range(n)]
m = E.MetricRadonIntermediate(ctx, Ps, dtrs)
P = E.pack_projection_matrices(Ps)
P2 = P.copy()
P2.reshape(-1)[200 * 12 + 9] += 1e-3  # view 200 moved a little: the optimiser's step
N = n * (n - 1) // 2
out = []
ref = {}
for mode in modes:
    m.debugSetXcdSchedule(mode)
    row = {"mode": mode}
    for world in worlds:
        first, count = (0, N) if world == 1 else sharding.balanced_pair_range(m, world // 2, world)
        for k in range(12):
            m.setProjectionMatrices(P2 if k & 1 else P); v = m.evaluate_range(first, count)
        ctx.enable_timing(True)
        ks = []
        for k in range(40):
            m.setProjectionMatrices(P2 if k & 1 else P); v = m.evaluate_range(first, count)
            ks.append(ctx.last_kernel_ms("pairs"))
        ctx.enable_timing(False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        steps = 300
        for k in range(steps):
            m.setProjectionMatrices(P2 if k & 1 else P); v = m.evaluate_range(first, count)
        torch.cuda.synchronize()
        step_us = 1e6 * (time.perf_counter() - t0) / steps
        key = (world, (steps - 1) & 1)
        assert ref.setdefault(key, v) == v, "the schedule changed a value"
        row["world%d" % world] = dict(pairs=count, kernel_us=1e3 * float(np.median(ks)), step_us=step_us)
    out.append(row)
    sys.stderr.write(json.dumps(row) + "\n")
print(json.dumps(out, indent=1))
