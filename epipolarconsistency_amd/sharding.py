"""Pair-space sharding for one process per GPU.

Pairs are independent; the only cross-pair operation of the path is the final sum
(ref: EpipolarConsistencyRadonIntermediate.cpp:216-224).  Every rank holds the whole (replicated)
dtr stack, evaluates a contiguous, equal-count chunk of the get_ij order and the ranks exchange one
float64 partial sum per evaluation (RCCL all-reduce on GPUs; gloo in the CPU tests).
"""


def pair_range(rank, world, n_pairs):
    """Contiguous shard [first, first+count) of the pair index range for `rank` of `world`."""
    first = rank * n_pairs // world
    return first, (rank + 1) * n_pairs // world - first


def view_range(rank, world, n_views):
    """Views whose Radon intermediates `rank` computes before the all-gather (equal chunks, the
    last ranks may get fewer)."""
    chunk = (n_views + world - 1) // world
    lo = min(rank * chunk, n_views)
    return lo, min(lo + chunk, n_views), chunk


def allreduce_mean(partial_sum_tensor, n_pairs, group=None):
    """In-place all-reduce(sum) of a 1-element float64 tensor holding this rank's partial sum;
    returns the mean over all pairs as a Python float."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if partial_sum_tensor.is_cuda and dist.get_backend(group) == "gloo":
            host = partial_sum_tensor.cpu()  # gloo rehearsal of the GPU path: reduce through host memory
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            return float(host.item()) / n_pairs
        dist.all_reduce(partial_sum_tensor, op=dist.ReduceOp.SUM, group=group)
    return float(partial_sum_tensor.item()) / n_pairs


def distributed_evaluate(metric, n_views, sum_tensor, rank, world, group=None):
    """One all-pairs evaluation sharded over `world` ranks: launches this rank's shard
    asynchronously on the metric's stream, all-reduces the 8-byte partial sum, returns the mean."""
    n_pairs = n_views * (n_views - 1) // 2
    first, count = pair_range(rank, world, n_pairs)
    metric.evaluate_range_async(first, count, sum_tensor)
    return allreduce_mean(sum_tensor, n_pairs, group)
