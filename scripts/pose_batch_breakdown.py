"""Where the time of a batched pose call goes (csrc/ecc_poses.hip): BASELINE config 5's 600 poses of view n/2 on the benchmark's
data set (400 views of 1024^2, 768^2 bins; dtrs = synthetic noise: the kernels' time does not depend on the values), the call
alone timed on the host (arrays prepared outside), K = 600 / 100 / 12 / 1, sparse and dense forms.  Run under
`rocprofv3 --kernel-trace --stats` for the device side."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic, geometry

n, S, B = 400, 1024, 768
if len(sys.argv) > 1:
    n = int(sys.argv[1])
Ks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [600, 100, 12, 1]
forms = sys.argv[3].split(",") if len(sys.argv) > 3 else ["deltas", "dense"]
ctx = E.Context(0)
rng = np.random.default_rng(0)
base_dtrs = [E.RadonIntermediate.from_host(ctx, rng.standard_normal((B, B)).astype(np.float32), S, S) for _ in range(8)]
dtrs = [base_dtrs[v % 8] for v in range(n)]
Ps = synthetic.short_scan(n, S, S, 0.308)
packed = E.pack_projection_matrices(Ps)
m = E.MetricRadonIntermediate(ctx, Ps, dtrs)
moving = n // 2
names = ["tx", "ty", "tz", "rx", "ry", "rz"]
ranges = [5.0, 5.0, 5.0] + [float(np.deg2rad(2.0))] * 3
rows = []
for q in range(600):
    p, k = divmod(q, 100)
    x = -ranges[p] + 2 * ranges[p] * k / 99.0
    rows.append((Ps[moving] @ geometry.rigid_transform(**{names[p]: x})).T.reshape(12))
rows = np.ascontiguousarray(np.stack(rows))
m.evaluate()
for K in Ks:
    off = np.arange(K + 1, dtype=np.int32)
    views = np.full(K, moving, np.int32)
    dense = np.repeat(packed[None], K, axis=0)
    dense[:, moving, :] = rows[:K]
    dense = np.ascontiguousarray(dense)
    for form in forms:
        ts = []
        for rep in range(6):
            m.setProjectionMatrices(packed)
            ctx.synchronize()
            t0 = time.perf_counter()
            if form == "deltas":
                got = m.evaluate_pose_deltas_packed(off, views, rows[:K])
            else:
                got = m.evaluate_poses(dense)
            ts.append(time.perf_counter() - t0)
        print("K = %3d %-6s  best %8.1f us  median %8.1f us  -> %9.0f evaluations/s   (batched %d)"
              % (K, form, 1e6 * min(ts), 1e6 * float(np.median(ts)), K / float(np.median(ts)), m.last_batched_poses()), flush=True)
