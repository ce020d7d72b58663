"""Child process of tests/test_gpu_rccl_one_rank.py: a ONE-rank RCCL process group on cuda:0 and the library's sharded
evaluation path over it, compared bit for bit with the plain single-device calls.

The exchange the path needs across GPUs is the sum of the pairs' values (ref:
EpipolarConsistencyRadonIntermediate.cpp:216-224).  With one rank the collective adds nothing, but everything around it is
the real thing: RCCL communicator set-up, the all-reduce / all-gather calls on torch's current stream (= the context's
stream), publish_scalar_kernel queued behind the collective, the poll of the pinned result slot.

Runs in its own process so that a failing or hanging RCCL cannot take the test session with it.  Prints one JSON line."""
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist

    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import sharding, synthetic

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "cases": []}

    probe = torch.tensor([41.0], dtype=torch.float64, device=dev)
    dist.all_reduce(probe)
    out["probe"] = probe.item()

    torch.cuda.set_stream(torch.cuda.Stream(dev))
    stream = torch.cuda.current_stream()
    ctx = E.Context(0, stream=stream.cuda_stream)
    # the library's own communicator (ecc_comm_*: ncclCommInitRank by the library, the id handed round by torch.distributed)
    comm = sharding.RcclComm(ctx, 0, 1, sharding.torch_broadcast_bytes(dev))
    # (views, image size, bins): 66 pairs (one-launch path of the plain call), 780 pairs, 4950 pairs (two-wave split kernel off)
    for n, S, B, mode in ((12, 96, 64, "auto"), (40, 128, 96, "auto"), (100, 128, 96, "polynomial")):
        Ps = synthetic.short_scan(n, S, S, 0.308 * 1024.0 / S)
        imgs = synthetic.projections_torch(Ps, S, S, synthetic.sphere_phantom(extent_mm=30, rmin=8, rmax=25), dev)
        slab = E.slab_floats(B, B)
        local = torch.zeros((n, slab), dtype=torch.float32, device=dev)
        keep = E.RadonIntermediate.compute_into(ctx, imgs, local, B, B)
        ctx.synchronize()
        # the Radon-intermediate stack through the collective bench.py uses for it
        gathered = torch.empty_like(local)
        dist.all_gather_into_tensor(gathered, local.contiguous())
        torch.cuda.synchronize()
        same_stack = bool(torch.equal(gathered, local))
        dtrs = [E.RadonIntermediate.wrap_device(ctx, gathered[k], B, B, S, S) for k in range(n)]
        metric = E.MetricRadonIntermediate(ctx, Ps, dtrs).setSampling(mode)
        n_pairs = n * (n - 1) // 2
        cost_plain = np.zeros((n, n), np.float32)
        want = metric.evaluate(cost_plain)
        sum_t = torch.zeros(1, dtype=torch.float64, device=dev)
        got_publish = sharding.distributed_evaluate(metric, n, sum_t, 0, 1, publish=True)
        got_item = sharding.distributed_evaluate(metric, n, sum_t, 0, 1, publish=False)
        # a moved view in between: the record-reuse path in front of the collective
        P2 = [p.copy() for p in Ps]
        P2[n // 2][:, 3] += 0.01 * P2[n // 2][:, 0]
        metric.setProjectionMatrices(P2)
        want2 = metric.evaluate()
        metric.setProjectionMatrices(Ps)
        metric.evaluate()
        metric.setProjectionMatrices(P2)
        got2 = sharding.distributed_evaluate(metric, n, sum_t, 0, 1, publish=True)
        # the same two evaluations with the all-reduce issued by the library (ecc_metric_evaluate_range_allreduce)
        got2_native = sharding.rccl_evaluate(metric, n, comm)
        metric.setProjectionMatrices(Ps)
        got_native = sharding.rccl_evaluate(metric, n, comm)
        half = n_pairs // 2  # and as two shards of one rank's range: the parts add up to the whole sum
        parts = metric.evaluate_range_allreduce(comm, 0, half) + metric.evaluate_range_allreduce(comm, half, n_pairs - half)
        metric.setProjectionMatrices(P2)
        # allreduce_mean on a tensor the caller filled, through the metric's result slot
        t = torch.tensor([want2 * n_pairs], dtype=torch.float64, device=dev)
        got3 = sharding.allreduce_mean(t, n_pairs, metric=metric)
        # cost image of a sharded evaluation through the all-gather
        metric.setProjectionMatrices(Ps)
        s, vals = metric.evaluate_range(0, n_pairs, want_pairs=True)
        cost = sharding.gather_cost_image(vals, n, 0, 1)
        out["cases"].append({"n": n, "pairs": n_pairs, "mode": mode, "same_stack": same_stack,
                             "want": want, "publish": got_publish, "item": got_item, "want_moved": want2, "publish_moved": got2,
                             "allreduce_mean": got3, "range_sum_over_pairs": s / n_pairs,
                             "native": got_native, "native_moved": got2_native, "native_parts_mean": parts / n_pairs,
                             "cost_image_equal": bool(np.array_equal(cost, cost_plain)),
                             "cost_image_nonzero": int(np.count_nonzero(cost))})
        metric.close()
        for d in dtrs:
            d.close()
        del keep
    comm.close()
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
