#!/usr/bin/env python3
"""bench.py -- ECC evaluations/s on the BASELINE workload (400 projections, 1024x1024, 768x768 Radon bins).

One "step" = one all-pairs evaluation as an optimiser iteration pays for it
(ref: Gui/SingleImageMotion.h:84-90 -> MetricRadonIntermediate::setProjectionMatrices + evaluate):
new projection matrices come from the host, the per-view pre-compute + upload, the pair kernel over
all n(n-1)/2 pairs, the float64 reduction and the scalar back on the host.  Radon intermediates are
resident in HBM when the timed region starts (they are computed once per data set, before it, and
that cost is reported separately as ms_per_radon_intermediate).

N GPUs: one process per GPU (torchrun), dtr stack produced data-parallel + all-gathered once (RCCL),
contiguous shards of the pair range per rank, and per evaluation the 8-byte partial sums -- already
on the host -- are added through the library's shared-memory exchange (--exchange collective: an
all-reduce of a device scalar instead).  The total work per evaluation is fixed, so scaling is "strong".

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
ENGINE_CLOCK_GHZ = 2.4  # MI355X peak engine clock (same guide); the pair kernel runs at ~2.1 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--views", type=int, default=400)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--bins", type=int, default=768)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo only to rehearse the multi-rank path)")
    ap.add_argument("--exchange", default="shm", choices=["shm", "collective"],
                    help="N > 1: per-evaluation sum of the partial results through the library's shared-memory "
                         "exchange (default; falls back to the collective if /dev/shm is unusable) or an all-reduce")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--cpu-sample-stride", type=int, default=1,
                    help="CPU baseline evaluates all pairs among every k-th view")
    return ap.parse_args()


def n_kappa_auto(n_u, n_v, n_t):
    """#{k >= 0 : dkappa*(k+1/2) < kappa_max} with dkappa = 2*kappa_max/num_samples,
    num_samples = n_t*step_t*2 (ref: ...RadonIntermediate.cu:320,337; .cu:257-263) = ceil(D - 1/2)."""
    import numpy as np
    step_t = np.float32(np.sqrt(float(n_v) ** 2 + float(n_u) ** 2) / n_t)
    D = float(np.float32(n_t) * step_t)
    return int(np.ceil(D - 0.5))


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import geometry, sharding, synthetic

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    n, S, B = args.views, args.size, args.bins
    pixel_mm = 0.308 * 1024.0 / S
    Ps = synthetic.short_scan(n, S, S, pixel_mm)
    phantom = synthetic.sphere_phantom()

    torch.cuda.set_stream(torch.cuda.Stream(dev))  # a stream of our own, not the legacy default stream
    stream = torch.cuda.current_stream()
    ctx = E.Context(local_rank, stream=stream.cuda_stream)
    ctx.enable_timing(True)  # Radon / pre-processing kernel times below

    # ---- Radon intermediates: data-parallel over views, then one all-gather ---------------------
    slab = E.slab_floats(B, B)
    chunk = (n + world - 1) // world
    lo, hi = min(rank * chunk, n), min((rank + 1) * chunk, n)
    slabs_all = torch.zeros((chunk * world, slab), dtype=torch.float32, device=dev)
    local = slabs_all[rank * chunk:(rank + 1) * chunk]
    radon_ms = 0.0
    sub = 50  # images generated and transformed 50 at a time (200 MB of projections in flight)
    for a in range(lo, hi, sub):
        b = min(a + sub, hi)
        imgs = synthetic.projections_torch(Ps[a:b], S, S, phantom, dev)
        keep = E.RadonIntermediate.compute_into(ctx, imgs, local[a - lo:b - lo], B, B)
        ctx.synchronize()
        radon_ms += ctx.last_kernel_ms("radon")
        del keep, imgs
    ms_per_radon = radon_ms / max(hi - lo, 1)
    # pre-processing (the step in front of the Radon intermediate) on one sub-batch, device
    # resident in and out, reference defaults + cosine weighting: HBM-bound, 8 B per pixel algorithmic
    imgs = synthetic.projections_torch(Ps[lo:min(lo + sub, hi)], S, S, phantom, dev)
    pre_out = torch.empty_like(imgs)
    pp = E.PreProccess()
    for _ in range(2):
        pp.process(ctx, imgs, Ps[lo:min(lo + sub, hi)], out=pre_out)
    ms_per_preprocess = ctx.last_kernel_ms("preprocess") / imgs.shape[0]
    del imgs, pre_out
    if world > 1:
        if args.backend == "nccl":
            gathered = torch.empty_like(slabs_all)
            dist.all_gather_into_tensor(gathered, local.contiguous())
        else:  # gloo rehearsal: through host memory
            parts = [torch.empty(local.shape, dtype=torch.float32) for _ in range(world)]
            dist.all_gather(parts, local.cpu())
            gathered = torch.cat(parts).to(dev)
        slabs_all = gathered
    dtrs = [E.RadonIntermediate.wrap_device(ctx, slabs_all[k], B, B, S, S) for k in range(n)]
    metric = E.MetricRadonIntermediate(ctx, Ps, dtrs)

    # ---- shard of the pair range --------------------------------------------------------------
    n_pairs = n * (n - 1) // 2
    first, count = sharding.pair_range(rank, world, n_pairs)
    sum_t = torch.zeros(1, dtype=torch.float64, device=dev)
    moving = n // 2  # view perturbed per step, like SingleImageMotion does for its input view

    P_pack = E.pack_projection_matrices(Ps)  # (n, 12) float64, what Eigen's Ps[i].data() holds
    P_moving = Ps[moving].copy()
    # what the optimiser hands over: view `moving` perturbed by a small rigid motion, 350 distinct poses prepared
    # ahead (producing a pose is the optimiser's work, not the metric's); the handover itself -- 38 KB of float64
    # into the library, the per-view pre-compute -- is inside every step
    poses = []
    for k in range(350):
        T = geometry.rigid_transform(tx=0.01 * (k % 50), rz=1e-4 * (k % 7))
        Pk = P_pack.copy()
        Pk[moving] = (P_moving @ T).T.reshape(12)
        poses.append(Pk)

    # N > 1: the per-evaluation exchange of the 8-byte partial sums
    exchange = None
    if world > 1 and args.exchange == "shm":
        ok = True
        try:
            exchange = sharding.open_exchange(rank, world, dist.barrier)
        except Exception as e:  # no usable /dev/shm
            sys.stderr.write("rank %d: shared-memory exchange unavailable (%s)\n" % (rank, e))
            exchange, ok = None, False
            if rank == 0:
                dist.barrier()  # the one open_exchange did not reach
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if flag.item() == 0:
            exchange = None  # all ranks take the collective
        elif abs(exchange.sum(float(rank + 1)) - world * (world + 1) / 2.0) > 1e-12:
            raise SystemExit("shared-memory exchange returned a wrong sum")

    def step(k):
        metric.setProjectionMatrices(poses[k % len(poses)])
        if world == 1:
            return metric.evaluate()
        if exchange is not None:
            return sharding.exchanged_evaluate(metric, n, exchange)
        return sharding.distributed_evaluate(metric, n, sum_t, rank, world)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ctx.enable_timing(False)  # no event records inside the timed region (they break back-to-back dispatch)
    for k in range(args.warmup):
        step(k)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        last = step(k)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        e = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        elapsed = e.item()
    # pair-kernel duration: HIP events on the context's stream around the pair kernel alone, averaged over a
    # second, untimed pass over the same steps (at most 50)
    ctx.enable_timing(True)
    pair_ms, n_timed = 0.0, max(1, min(args.steps, 50))
    for k in range(n_timed):
        step(k)
        pair_ms += ctx.last_kernel_ms("pairs")
    fence()
    pair_ms /= n_timed

    n_kappa = n_kappa_auto(S, S, B)
    bytes_per_pair = 64 * n_kappa + 68            # 2 views x 2 signs x 4 taps x 4 B + K01 + result
    launch_bytes = bytes_per_pair * count         # one launch = this rank's shard of pairs
    achieved = launch_bytes / (pair_ms * 1e-3) / 1e9 if pair_ms > 0 else 0.0

    out = {
        "metric": "ECC evaluations/sec (N=%d, %d^2 projections)" % (n, S),
        "value": args.steps / elapsed,
        "unit": "evaluations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "%d-projection %dx%d circular short scan, %dx%d Radon bins, all %d pairs"
                               % (n, S, S, B, B, n_pairs),
                   "n_kappa_per_pair": n_kappa, "pairs_per_rank": count, "parallelism": "pair-shard x%d" % world,
                   "sum_exchange": "none" if world == 1 else ("host shared memory" if exchange is not None
                                                               else "all-reduce (%s)" % args.backend)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "pairs_kernel<true, false>", "kernel_ms": pair_ms,
                     "algorithmic_bytes_per_launch": launch_bytes},
        "ms_per_radon_intermediate": ms_per_radon,
        "ms_per_preprocess": ms_per_preprocess,
        "preprocess_hbm_GBs": 8.0 * S * S / (ms_per_preprocess * 1e-3) / 1e9 if ms_per_preprocess > 0 else 0.0,
        "pairs_per_s": n_pairs * args.steps / elapsed,
        "kappa_samples_per_s": n_pairs * n_kappa * args.steps / elapsed,
        "last_value": last,
    }

    # The gather is served on chip (traffic << algorithmic bytes), so the bandwidth that actually bounds the kernel is
    # the L1's: 64 B/clk/CU (scripts/micro/gather_rate.hip); every algorithmic byte passes through it exactly once.
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    l1_peak = 64.0 * n_cu * ENGINE_CLOCK_GHZ  # GB/s
    out["roofline"]["l1"] = {"achieved": achieved, "peak": l1_peak, "unit": "GB/s", "frac": achieved / l1_peak,
                             "note": "64 B/clk/CU x %d CUs x %.1f GHz peak engine clock" % (n_cu, ENGINE_CLOCK_GHZ)}

    # measured HBM traffic per launch (rocprofv3 PMC pass, committed under profiles/), if it matches
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tfile):
        try:
            t = json.load(open(tfile))
            if t.get("views") == n and t.get("size") == S and t.get("bins") == B and t.get("n_gpus", 1) == world:
                out["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
        except Exception:
            pass

    # ---- CPU baseline: the oracle timed on this box's host cores (rank 0, N = 1 only) -----------
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        import oracle
        oracle.build(native=True)
        views = list(range(0, n, args.cpu_sample_stride))
        host_dtrs = [dtrs[v].readback() for v in views]
        Psub = [Ps[v] for v in views]
        oracle.evaluate_all(Psub[:8], host_dtrs[:8], S, S, native=True)  # warm-up (thread pool, page-in)
        reps, cpu_s, ref = 0, 0.0, None
        while reps < 3 or (cpu_s < 12.0 and reps < 60):  # bounded: 10-15 s of CPU work
            t1 = time.perf_counter()
            ref = oracle.evaluate_all(Psub, host_dtrs, S, S, native=True)
            cpu_s += time.perf_counter() - t1
            reps += 1
        sub_pairs = len(views) * (len(views) - 1) // 2
        cpu_evals = (sub_pairs * reps / cpu_s) / n_pairs
        # the same sub-problem on the GPU doubles as a full-size parity check
        metric.setProjectionMatrices(Ps)
        gpu_mean = metric.evaluate(set(views)) if args.cpu_sample_stride > 1 else metric.evaluate()
        out["cpu_baseline"] = {
            "value": cpu_evals, "unit": "evaluations/s", "cores": oracle.lib().eccor_num_threads(), "kind": "port",
            "sample": "%d x all %d pairs among every %d-th view (%d of %d pairs per evaluation) in %.1f s, "
                      "scaled by pair count; oracle/ecc_oracle.c -O3 -march=native -fopenmp"
                      % (reps, sub_pairs, args.cpu_sample_stride, sub_pairs, n_pairs, cpu_s),
        }
        out["parity_rel_err_vs_oracle_on_sample"] = abs(gpu_mean - ref["mean"]) / abs(ref["mean"])
        # Radon baseline: 1/4 of the bins of one image
        img = synthetic.projections_numpy([Ps[n // 3]], S, S, phantom)[0]
        bins = np.arange(0, B * B, 4, dtype=np.int32)
        t1 = time.perf_counter()
        oracle.radon_bins(img, B, B, bins, native=True)
        out["cpu_baseline"]["ms_per_radon_intermediate"] = 1e3 * (time.perf_counter() - t1) * 4

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        if exchange is not None:
            exchange.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
