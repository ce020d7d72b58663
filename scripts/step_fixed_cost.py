#!/usr/bin/env python3
"""Where the fixed cost of one optimiser step goes (GPU box): python scripts/step_fixed_cost.py [world] [steps]
One view moves per step (ref: Gui/SingleImageMotion.h:84-90), like bench.py.  world = 1: all 79 800 pairs (the step bench.py
times); world = G: the middle rank's cost-balanced shard of a G-rank job (scripts/shard_step.py's step, but WITH the moved
view).  Reports the step time, the pair kernel's time (HIP events, separate pass), and the host's stamps inside the two
calls (ecc_debug_step_stamps): set_projections, change detection, first launch, refit launches, sum launch, result."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import _lib, geometry, sharding, synthetic

world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
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
P_pack = E.pack_projection_matrices(Ps)
moving = n // 2
poses = []
for k in range(64):
    Pk = P_pack.copy()
    Pk[moving] = (Ps[moving] @ geometry.rigid_transform(tx=0.01 * (k % 50), rz=1e-4 * (k % 7))).T.reshape(12)
    poses.append(Pk)
n_pairs = n * (n - 1) // 2
first, count = (0, n_pairs) if world == 1 else sharding.balanced_pair_range(metric, world // 2, world)


def step(k):
    metric.setProjectionMatrices(poses[k % len(poses)])
    return metric.evaluate() if world == 1 else metric.evaluate_range(first, count)


stamps = np.zeros(8)
for reuse in (True, False):
    metric.setRecordReuse(reuse)
    for k in range(40):
        step(k)
    ctx.enable_timing(False)
    torch.cuda.synchronize()
    acc = np.zeros(8)
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
        _lib.lib().ecc_debug_step_stamps(metric._h, C.c_void_p(stamps.ctypes.data))
        acc += stamps - stamps[0]
    el = (time.perf_counter() - t0) / steps
    ctx.enable_timing(True)
    pk = 0.0
    for k in range(100):
        step(k)
        pk += ctx.last_kernel_ms("pairs")
    pk = pk / 100 * 1e3
    a = acc / steps * 1e6
    print("world %d (%d pairs), record reuse %s: step %.1f us, pair kernel %.1f us, non-pair %.1f us" % (world, count, reuse, el * 1e6, pk, el * 1e6 - pk))
    print("   host stamps (us after set_projections entered): set returned %.1f | evaluate entered %.1f | change detection done %.1f | "
          "first pair launch returned %.1f | refit launches queued %.1f | sum launch returned %.1f | result seen %.1f | "
          "(next set_projections %.1f later)" % (a[1], a[2], a[3], a[4], a[5], a[6], a[7], el * 1e6 - a[7]))
