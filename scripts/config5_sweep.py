"""BASELINE config 5: FDCTMotionCorrection-style inner loop.  View n/2 of the 400-view scan is
perturbed by the "3D Rigid" parameters (ref: LibProjectiveGeometry/Models/ModelSimilarity3D.hxx:64-88,
P' = P T) and swept like plotCostFunction does (ref: Gui/Visualization.h:78-98): 6 parameters x 100
steps over [-5, 5] mm / [-2, 2] deg = 600 full all-pairs evaluations.  Reports evaluations/s and checks
6 sampled sweep points against the CPU oracle.

One GPU:   python scripts/config5_sweep.py [n size bins]
N GPUs:    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/config5_sweep.py
           -- the 600 sweep points are independent evaluations: rank r takes every N-th point, every rank holds the
           whole dtr stack, nothing is exchanged inside the timed loop ("shard the sweep points"); the
           values are gathered afterwards (gloo).  ECC_SWEEP_SINGLE_DEVICE=1 puts all ranks on cuda:0 (rehearsal).
ECC_SWEEP_INCREMENTAL=1 (one GPU): the sweep is run a second time with ecc_metric_set_incremental -- only the 399 pairs of
the moving view are re-evaluated per point -- and the 600 values are compared bit for bit with the full evaluations'."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import geometry, synthetic

n, S, B = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (400, 1024, 768)
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
local = 0 if os.environ.get("ECC_SWEEP_SINGLE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("gloo")
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
torch.cuda.set_stream(torch.cuda.Stream(dev))
Ps = synthetic.short_scan(n, S, S, 0.308 * 1024 / S)
ph = synthetic.sphere_phantom()
ctx = E.Context(local, stream=torch.cuda.current_stream().cuda_stream)
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
values = np.zeros(600)
mine = range(rank, 600, world)
if world > 1:
    dist.barrier()
t0 = time.perf_counter()
for q in mine:
    p, k = divmod(q, 100)
    x = -ranges[p] + 2 * ranges[p] * k / 99.0
    packed[moving] = (P0 @ geometry.rigid_transform(**{names[p]: x})).T.reshape(12)
    m.setProjectionMatrices(packed)
    values[q] = m.evaluate()
elapsed = time.perf_counter() - t0
if world > 1:
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    v = torch.from_numpy(values)
    dist.all_reduce(v, op=dist.ReduceOp.SUM)  # every point was evaluated by exactly one rank
    values = v.numpy()
values = values.reshape(6, 100)
out = {"config": "config 5: %d views %dx%d, view %d swept over 6 rigid parameters x 100 steps" % (n, S, S, moving),
       "evaluations": 600, "seconds": elapsed, "evaluations_per_s": 600 / elapsed, "n_gpus": world,
       "decomposition": "sweep points round-robin over ranks, no exchange in the timed loop" if world > 1 else "one GPU",
       "min_at_step": [int(np.argmin(values[p])) for p in range(6)]}
if world == 1 and os.environ.get("ECC_SWEEP_INCREMENTAL"):
    m.setIncremental(True)
    packed[moving] = P0.T.reshape(12)
    m.setProjectionMatrices(packed).evaluate()  # the kept values: one full evaluation, outside the timed loop like the warm-up
    inc_values = np.zeros(600)
    recomputed = set()
    t0 = time.perf_counter()
    for q in range(600):
        p, k = divmod(q, 100)
        x = -ranges[p] + 2 * ranges[p] * k / 99.0
        packed[moving] = (P0 @ geometry.rigid_transform(**{names[p]: x})).T.reshape(12)
        m.setProjectionMatrices(packed)
        inc_values[q] = m.evaluate()
        recomputed.add(m.last_evaluated_pairs())
    inc_elapsed = time.perf_counter() - t0
    m.setIncremental(False)
    out["incremental"] = {"seconds": inc_elapsed, "evaluations_per_s": 600 / inc_elapsed,
                          "pairs_recomputed_per_evaluation": sorted(recomputed),
                          "values_bit_identical_to_full_evaluations": bool(np.array_equal(inc_values.reshape(6, 100), values))}
if rank == 0:
    # oracle spot checks (6 of the 600 points)
    import oracle
    oracle.build(native=True)
    host = [d.readback() for d in dtrs]
    errs = []
    for p, k in ((0, 10), (1, 90), (2, 49), (3, 0), (4, 70), (5, 99)):
        x = -ranges[p] + 2 * ranges[p] * k / 99.0
        Pk = [q.copy() for q in Ps]
        Pk[moving] = P0 @ geometry.rigid_transform(**{names[p]: x})
        ref = oracle.evaluate_all(Pk, host, S, S, native=True)["mean"]
        errs.append(abs(values[p, k] - ref) / abs(ref))
    out["parity_rel_err_at_6_points"] = errs
    print(json.dumps(out))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
