"""MetricDirect micro-bench (GPU box): seconds per all-pairs evaluation and line integrals/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda", 0)
Ps = synthetic.short_scan(n, S, S, 0.308 * 1024 / S)
imgs = synthetic.projections_torch(Ps, S, S, synthetic.sphere_phantom(), dev)
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
m = E.MetricDirect(ctx, Ps, imgs)
m.evaluate()
t0 = time.perf_counter(); reps = 3
for _ in range(reps):
    v = m.evaluate()
dt = (time.perf_counter() - t0) / reps
pairs = n * (n - 1) // 2
lines = pairs * 2 * int(2 * np.sqrt(2.0) * S)
print("n=%d %dx%d: %.2f ms per all-pairs evaluation (%d pairs, %.3g line integrals, %.3g lines/s), sum %.6g"
      % (n, S, S, 1e3 * dt, pairs, lines, lines / dt, v))
