#!/usr/bin/env python3
"""Times one rank's share of a sharded evaluation on one GPU: python scripts/shard_step.py [world] [steps].
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel durations at shard size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import sharding, synthetic

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n, S, B = 400, 1024, 768
Ps = synthetic.short_scan(n, S, S, 0.308)
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(dev))
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
ph = synthetic.sphere_phantom()
for a in range(0, n, 50):
    imgs = synthetic.projections_torch(Ps[a:a + 50], S, S, ph, dev)
    keep = E.RadonIntermediate.compute_into(ctx, imgs, slabs[a:a + 50], B, B)
    ctx.synchronize()
dtrs = [E.RadonIntermediate.wrap_device(ctx, slabs[k], B, B, S, S) for k in range(n)]
metric = E.MetricRadonIntermediate(ctx, Ps, dtrs)
P = E.pack_projection_matrices(Ps)
ranks = [int(r) for r in sys.argv[3].split(",")] if len(sys.argv) > 3 else [world // 2]
for rank in ranks:
    if os.environ.get("ECC_EQUAL_COUNT_SHARDS"):
        first, count = sharding.pair_range(rank, world, n * (n - 1) // 2)
    else:
        first, count = sharding.balanced_pair_range(metric, rank, world)
    for _ in range(20):
        metric.setProjectionMatrices(P); metric.evaluate_range(first, count)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        metric.setProjectionMatrices(P); metric.evaluate_range(first, count)
    torch.cuda.synchronize()
    print("world %d rank %d: %d pairs, %.1f us per step (one rank, no exchange)" % (world, rank, count, 1e6 * (time.perf_counter() - t0) / steps))
