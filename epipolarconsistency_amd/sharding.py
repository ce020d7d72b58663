"""Pair-space sharding for one process per GPU.

Pairs are independent; the only cross-pair operation of the path is the final sum
(ref: EpipolarConsistencyRadonIntermediate.cpp:216-224).  Every rank holds the whole (replicated)
dtr stack, evaluates a contiguous, equal-count chunk of the get_ij order and the ranks exchange one
float64 partial sum per evaluation: through the library's host-side exchange (ScalarExchange -- the partial sums
are on the host already, a shared-memory cache line per rank costs ~1 us) or, with --exchange rccl, an RCCL
all-reduce of a device scalar (gloo in the CPU tests).  Bulk data (the dtr stack) always moves with RCCL.
"""
import ctypes as C
import os


def pair_range(rank, world, n_pairs):
    """Contiguous shard [first, first+count) of the pair index range for `rank` of `world`."""
    first = rank * n_pairs // world
    return first, (rank + 1) * n_pairs // world - first


def balanced_pair_range(metric, rank, world):
    """Cost-balanced contiguous shard of `rank` (ecc_metric_balanced_shards): the pair kernel's time per pair grows
    with the pair's kappa_max, and for a circular scan the expensive pairs sit in the first rows of the pair triangle,
    so equal-count shards leave rank 0 the straggler (8 ranks: 93 us against 68 us).  Every rank computes the same
    boundaries from the same matrices.  Call once per data set and keep the result for all evaluations."""
    b = metric.balanced_shards(world)
    return b[rank], b[rank + 1] - b[rank]


def view_range(rank, world, n_views):
    """Views whose Radon intermediates `rank` computes before the all-gather (equal chunks, the
    last ranks may get fewer)."""
    chunk = (n_views + world - 1) // world
    lo = min(rank * chunk, n_views)
    return lo, min(lo + chunk, n_views), chunk


def allreduce_mean(partial_sum_tensor, n_pairs, group=None, metric=None):
    """In-place all-reduce(sum) of a 1-element float64 tensor holding this rank's partial sum;
    returns the mean over all pairs as a Python float.
    metric: when given (and the tensor lives on the device the metric's context works on, torch's current stream being the
    context's stream), the reduced value comes back through the metric's pinned result slot (a one-thread kernel queued
    behind the collective + a poll) instead of tensor.item() (a device-to-host copy command and its synchronisation)."""
    import torch.distributed as dist
    # whenever a process group exists the sum goes through its collective, a one-rank group included: that is the form in
    # which the RCCL path (all-reduce kernel -> publish_scalar_kernel, stream-ordered behind the pair kernel) can be run
    # and checked on a single GPU (tests/test_gpu_rccl_one_rank.py, bench.py --force-collective)
    if dist.is_available() and dist.is_initialized():
        if partial_sum_tensor.is_cuda and dist.get_backend(group) == "gloo":
            host = partial_sum_tensor.cpu()  # gloo rehearsal of the GPU path: reduce through host memory
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            return float(host.item()) / n_pairs
        dist.all_reduce(partial_sum_tensor, op=dist.ReduceOp.SUM, group=group)
    if metric is not None and partial_sum_tensor.is_cuda:
        metric.publish_scalar(partial_sum_tensor)
        return metric.wait_scalar() / n_pairs
    return float(partial_sum_tensor.item()) / n_pairs


def distributed_evaluate(metric, n_views, sum_tensor, rank, world, group=None, shard=None, publish=False):
    """One all-pairs evaluation sharded over `world` ranks: launches this rank's shard
    asynchronously on the metric's stream, all-reduces the 8-byte partial sum, returns the mean.
    shard: (first, count) of this rank (default: the equal-count chunk).  publish: see allreduce_mean(metric=...)."""
    n_pairs = n_views * (n_views - 1) // 2
    first, count = shard if shard is not None else pair_range(rank, world, n_pairs)
    metric.evaluate_range_async(first, count, sum_tensor)
    return allreduce_mean(sum_tensor, n_pairs, group, metric=metric if publish else None)


class ScalarExchange:
    """Sum of one float64 per rank between the processes of one node (ecc_exchange_* of the C ABI).
    Rank 0 creates the shared-memory segment, the others wait for it; every rank gets the same bits back
    (fixed summation order).  `name` must be the same on all ranks and unique per job."""

    def __init__(self, rank, world, name=None):
        from . import _lib
        from .api import check
        if name is None:
            name = default_exchange_name()
        self._h = C.c_void_p()
        self.rank, self.world, self.name = rank, world, name
        check(_lib.lib().ecc_exchange_open(name.encode(), int(rank), int(world), C.byref(self._h)))

    def sum(self, partial):
        from . import _lib
        from .api import check
        out = C.c_double()
        check(_lib.lib().ecc_exchange_sum(self._h, float(partial), C.byref(out)))
        return out.value

    def close(self):
        from . import _lib
        if self._h:
            _lib.lib().ecc_exchange_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def open_exchange(rank, world, barrier, name=None):
    """Opens the exchange on all ranks: rank 0 first (it removes a stale segment of the same name and creates a
    fresh one), then -- after `barrier()` -- the others, so that nobody can attach to a leftover of a crashed job.
    The segment is node-local shared memory: a job whose ranks span several nodes (WORLD_SIZE != LOCAL_WORLD_SIZE
    under torchrun) is refused here -- every rank raises before touching /dev/shm -- and takes the collective."""
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if local_world != world:
        raise RuntimeError("the shared-memory exchange is single-node (WORLD_SIZE %d, LOCAL_WORLD_SIZE %d)"
                           % (world, local_world))
    ex = ScalarExchange(rank, world, name) if rank == 0 else None
    barrier()
    if rank != 0:
        ex = ScalarExchange(rank, world, name)
    return ex


def default_exchange_name():
    """One name per torchrun job: the rendezvous port (same on all ranks) and the user id."""
    return "/ecc_hip_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getuid())


def exchanged_evaluate(metric, n_views, exchange, shard=None):
    """One all-pairs evaluation sharded over the ranks of `exchange`: this rank's shard through the synchronous
    range call (partial sum lands in pinned host memory), then the host-side sum; returns the mean.
    shard: (first, count) of this rank (default: the equal-count chunk)."""
    n_pairs = n_views * (n_views - 1) // 2
    first, count = shard if shard is not None else pair_range(exchange.rank, exchange.world, n_pairs)
    return exchange.sum(metric.evaluate_range(first, count)) / n_pairs


class RcclComm:
    """The library's own RCCL communicator of one rank (ecc_comm_* of the C ABI): rank 0 makes the 128-byte id, `broadcast`
    hands it to the others (a callable bytes -> bytes that every rank calls, e.g. over torch.distributed), then every rank joins
    (ncclCommInitRank on the context's device: a collective)."""

    def __init__(self, ctx, rank, world, broadcast, agree=None, timeout_s=None):
        """agree: callable int -> int that every rank calls, returning the MINIMUM over the ranks (torch_agree_min); with it the
        ranks settle whether all of them can bind RCCL BEFORE anybody enters ncclCommInitRank -- a rank that cannot would
        return at once and leave the others inside the collective for ever (advisor, round 5).
        timeout_s: ncclCommInitRank runs on a helper thread and this rank gives up after that many seconds (TimeoutError; the
        thread stays behind, `hung` is set: the caller should report and end the process with os._exit when it is done)."""
        from . import _lib
        from .api import check
        self._h = C.c_void_p()
        self.ctx, self.rank, self.world = ctx, rank, world
        self.hung = False
        L = _lib.lib()
        local_ok = L.ecc_comm_available() == 0
        why_local = "" if local_ok else L.ecc_last_error().decode(errors="replace")
        if agree is not None and int(agree(1 if local_ok else 0)) == 0:
            raise RuntimeError("RCCL cannot be bound on every rank" + (" (this rank: %s)" % why_local if why_local else " (this rank could)"))
        buf = (C.c_char * 128)()
        why = ""
        if rank == 0 and L.ecc_comm_unique_id(C.cast(buf, C.c_void_p)) != 0:
            why = L.ecc_last_error().decode(errors="replace")
            buf = (C.c_char * 128)()  # all zero: every rank learns that there is no id and raises, nobody waits for a collective
        ident = broadcast(bytes(buf))
        assert len(ident) == 128
        if not any(ident):
            raise RuntimeError("no RCCL communicator id from rank 0" + (": " + why if why else ""))
        if not local_ok:  # (without `agree`: this rank at least does not pretend)
            raise RuntimeError("RCCL cannot be bound on this rank: " + why_local)
        if timeout_s is None:
            check(L.ecc_comm_create(ctx._h, C.c_char_p(ident), int(rank), int(world), C.byref(self._h)))
            return
        import threading
        box = {}

        def run():
            rc = L.ecc_comm_create(ctx._h, C.c_char_p(ident), int(rank), int(world), C.byref(self._h))
            box["rc"] = rc
            box["err"] = L.ecc_last_error().decode(errors="replace") if rc else ""  # (the error text is thread-local)
        t = threading.Thread(target=run, daemon=True)
        t.start()
        t.join(float(timeout_s))
        if t.is_alive():
            self.hung = True
            raise TimeoutError("ncclCommInitRank did not return within %.0f s on rank %d of %d" % (timeout_s, rank, world))
        if box["rc"] != 0:
            from ._lib import EccError
            raise EccError(box["rc"], box["err"])

    def close(self):
        from . import _lib
        if self._h:
            _lib.lib().ecc_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def torch_broadcast_bytes(device=None, group=None):
    """-> broadcast(bytes) over the torch.distributed process group, from rank 0 (for RcclComm); the identity without a group"""
    def bc(b):
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return b
        t = torch.tensor(list(b), dtype=torch.uint8, device=device if (device is not None and dist.get_backend(group) == "nccl") else "cpu")
        dist.broadcast(t, 0, group=group)
        return bytes(t.cpu().tolist())
    return bc


def torch_agree_min(device=None, group=None):
    """-> agree(int) = the minimum over the ranks of the torch.distributed process group (for RcclComm); the identity without one"""
    def agree(v):
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return int(v)
        t = torch.tensor([int(v)], dtype=torch.int32, device=device if (device is not None and dist.get_backend(group) == "nccl") else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return int(t.item())
    return agree


def rccl_evaluate(metric, n_views, comm, shard=None):
    """One all-pairs evaluation sharded over the ranks of `comm` with the exchange inside the library call (pair kernel -> sum ->
    ncclAllReduce -> publish, one stream); returns the mean.  shard: (first, count) of this rank (default: the equal-count chunk)."""
    n_pairs = n_views * (n_views - 1) // 2
    first, count = shard if shard is not None else pair_range(comm.rank, comm.world, n_pairs)
    return metric.evaluate_range_allreduce(comm, first, count) / n_pairs


def gather_cost_image(pair_values, n_views, rank, world, group=None, cost=None):
    """The n x n cost image of a sharded evaluation ("when the caller wants the cost image, a gather of
    each rank's pair values"): every rank passes the float32 values of its pair_range shard (evaluate_range(...,
    want_pairs=True)), every rank gets the image back -- entry [j, i] = index i + j*n for i < j, other entries of
    `cost` untouched, like evaluate(cost) on one GPU.  Uses the process group (RCCL or gloo); not on the
    per-evaluation critical path of an optimiser, which only needs the scalar."""
    import numpy as np
    import torch
    import torch.distributed as dist
    n_pairs = n_views * (n_views - 1) // 2
    if cost is None:
        cost = np.zeros((n_views, n_views), np.float32)
    counts = [pair_range(r, world, n_pairs)[1] for r in range(world)]
    mine = torch.zeros(max(counts), dtype=torch.float32)
    mine[:counts[rank]] = torch.from_numpy(np.ascontiguousarray(pair_values, np.float32))
    # The collective only when the process group IS the sharding (its size equals `world`: a one-rank group too -- the RCCL
    # all-gather then runs on the one GPU there is).  world = 1 inside a larger group -- every rank evaluated all pairs --
    # stays local; any other mismatch is a caller error (an all_gather with the wrong list length errors or hangs).
    in_group = dist.is_available() and dist.is_initialized()
    if in_group and dist.get_world_size(group) != world:
        if world != 1:
            raise ValueError("gather_cost_image: world = %d but the process group has %d ranks" % (world, dist.get_world_size(group)))
        in_group = False
    if in_group:
        on_gpu = dist.get_backend(group) == "nccl"
        if on_gpu:
            mine = mine.cuda()
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        parts = [p.cpu() for p in parts]
    else:
        parts = [mine]
    vals = np.concatenate([parts[r][:counts[r]].numpy() for r in range(len(parts))])
    iu = np.triu_indices(n_views, 1)  # (i, j), i < j, in get_ij order
    cost[iu[1], iu[0]] = vals
    return cost
