"""Soak of the batched pose evaluation on the headline problem (GPU box): 400 views at 768 x 768 bins; batches of 1 ... 40 poses drawn
from a pool of 96 (one or two moved views each; bases alternate between the scan and two perturbed trajectories, so the kept base
values are redone by deltas now and then), in the delta form and the full-matrix form, with ordinary evaluations in between;
every mean compared with the one a metric without batching, record reuse and small-evaluation paths gave for the same matrices.
usage: soak_poses.py [batches, default 3000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic, geometry

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n, S, B = 400, 1024, 768
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
rng = np.random.default_rng(5)
bases = [P]
for b in range(2):
    Pb = P.copy()
    for v in (17 + 40 * b, 230 + b):
        Pb[v] = (Ps[v] @ geometry.rigid_transform(ty=0.3 + 0.1 * b, rx=2e-4)).T.reshape(12)
    bases.append(Pb)
pool = []  # (base, views, rows)
for k in range(96):
    b = k % 3
    vk = sorted({(n // 2, n // 3, 5, 399)[k % 4]} | ({int(rng.integers(1, n))} if k % 5 == 0 else set()))
    rows = np.stack([(bases[b][v].reshape(4, 3).T @ geometry.rigid_transform(tx=0.01 * (k % 16 + 1), rz=1e-4 * (k % 7))).T.reshape(12) for v in vk])
    pool.append((b, vk, rows))
ref = E.MetricRadonIntermediate(ctx, Ps, dtrs).setRecordReuse(False).setSmallEval(False).setPoseBatching(False)
want, want_base = [], [ref.setProjectionMatrices(Pb).evaluate() for Pb in bases]
for b, vk, rows in pool:
    Pk = bases[b].copy()
    Pk[vk] = rows
    want.append(ref.setProjectionMatrices(Pk).evaluate())
ref.close()
m = E.MetricRadonIntermediate(ctx, Ps, dtrs)
bad, poses_done, t0 = 0, 0, time.time()
for it in range(batches):
    b = int(rng.integers(0, 3))
    cand = [k for k in range(96) if pool[k][0] == b]
    sel = [cand[int(q)] for q in rng.integers(0, len(cand), size=int(rng.integers(1, 41)))]
    m.setProjectionMatrices(bases[b])
    if it % 2 == 0:
        got = m.evaluate_pose_deltas([pool[k][1] for k in sel], [pool[k][2] for k in sel])
    else:
        full = np.repeat(bases[b][None], len(sel), axis=0)
        for q, k in enumerate(sel):
            full[q][pool[k][1]] = pool[k][2]
        got = m.evaluate_poses(np.ascontiguousarray(full))
        m.setProjectionMatrices(bases[b])
    exp = np.array([want[k] for k in sel])
    # (the full-matrix form hands a single pose to the sequential path: nothing to batch)
    if not np.array_equal(got, exp) or m.last_batched_poses() != (0 if (it % 2 == 1 and len(sel) == 1) else len(sel)):
        bad += 1
        if bad < 10:
            print("MISMATCH batch %d (base %d, %d poses, batched %d): %s" % (it, b, len(sel), m.last_batched_poses(), np.flatnonzero(got != exp)[:5]), flush=True)
    if it % 7 == 0 and m.evaluate() != want_base[b]:
        bad += 1
        print("MISMATCH evaluate after batch %d" % it, flush=True)
    poses_done += len(sel)
    if (it + 1) % 1000 == 0:
        print("%d batches, %d poses, %d mismatches, %.1f s" % (it + 1, poses_done, bad, time.time() - t0), flush=True)
print("soak: %d batches, %d poses, %d mismatches, %.1f us per pose" % (batches, poses_done, bad, 1e6 * (time.time() - t0) / poses_done))
sys.exit(1 if bad else 0)
