"""Multi-rank path on CPU: world_size-2 gloo.  The data path has no collective except the 8-byte
all-reduce of partial sums, so the test exercises the shard arithmetic + all-reduce with the
oracle's pair values standing in for what each rank's GPU shard would produce."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, pairs, n_views, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from epipolarconsistency_amd import sharding
    n_pairs = n_views * (n_views - 1) // 2
    first, count = sharding.pair_range(rank, world, n_pairs)
    part = torch.tensor([float(np.sum(pairs[first:first + count].astype(np.float64)))], dtype=torch.float64)
    mean = sharding.allreduce_mean(part, n_pairs)
    cost = sharding.gather_cost_image(pairs[first:first + count], n_views, rank, world)
    out_q.put((rank, first, count, mean, cost))
    dist.destroy_process_group()


def test_pair_ranges_partition_everything():
    from epipolarconsistency_amd import sharding
    for n_pairs in (1, 7, 28, 79800):
        for world in (1, 2, 3, 8):
            got = [sharding.pair_range(r, world, n_pairs) for r in range(world)]
            assert got[0][0] == 0 and sum(c for _, c in got) == n_pairs
            for (f0, c0), (f1, _) in zip(got, got[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in got) - min(c for _, c in got) <= 1
    assert [sharding.view_range(r, 8, 400)[:2] for r in (0, 7)] == [(0, 50), (350, 400)]
    assert sharding.view_range(3, 4, 10) == (9, 10, 3)


@pytest.mark.timeout(120)
def test_two_rank_allreduce_mean(oracle_mod, small_scan):
    s = small_scan
    ref = oracle_mod.evaluate_all(s["Ps"], s["dtrs"], s["n_u"], s["n_v"])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ref["pairs"], 8, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=100) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert res[0][1:3] == (0, 14) and res[1][1:3] == (14, 14)
    for _, _, _, mean, cost in res:
        assert abs(mean - ref["mean"]) <= 1e-12 * abs(ref["mean"])
        assert np.array_equal(cost, ref["cost"])  # the gathered cost image of the shards == the single-process one


# ---- host-side exchange of the partial sums (ecc_exchange_* of the C ABI; no device involved) --------------------
def _exchange_worker(rank, world, port, pairs, n_views, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["ECC_NO_TORCH_PRELOAD"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from epipolarconsistency_amd import sharding
    ex = sharding.open_exchange(rank, world, dist.barrier)
    n_pairs = n_views * (n_views - 1) // 2
    first, count = sharding.pair_range(rank, world, n_pairs)
    part = float(np.sum(pairs[first:first + count].astype(np.float64)))
    means = [ex.sum(part * (1 + k)) / n_pairs for k in range(2000)]  # many generations: both slot sets, no lost update
    out_q.put((rank, means[0], means[1999], float(np.float64(means[1]) - 2 * np.float64(means[0]))))
    dist.barrier()
    ex.close()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 4])
def test_shared_memory_exchange(oracle_mod, small_scan, world):
    s = small_scan
    ref = oracle_mod.evaluate_all(s["Ps"], s["dtrs"], s["n_u"], s["n_v"])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, ref["pairs"], 8, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert len(set(r[1:] for r in res)) == 1, "all ranks must return identical bits"
    assert abs(res[0][1] - ref["mean"]) <= 1e-12 * abs(ref["mean"])
    assert abs(res[0][2] - 2000 * ref["mean"]) <= 1e-9 * abs(ref["mean"])


def test_exchange_times_out_instead_of_hanging(monkeypatch):
    """A rank that never shows up must not hang the others (a hung GPU job is worse than a failed one)."""
    import time
    monkeypatch.setenv("ECC_EXCHANGE_TIMEOUT_S", "0.3")
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import sharding
    ex = sharding.ScalarExchange(0, 2, "/ecc_hip_test_timeout_%d" % os.getpid())
    t0 = time.time()
    with pytest.raises(E.EccError) as ei:
        ex.sum(1.0)
    assert time.time() - t0 < 20 and "timed out" in str(ei.value)
    # the exchange stays failed: a retry returns at once instead of publishing the next generation
    t0 = time.time()
    with pytest.raises(E.EccError) as ei:
        ex.sum(1.0)
    assert time.time() - t0 < 0.2 and "failed earlier" in str(ei.value)
    ex.close()
    # single node only: refused before any segment is touched when the job spans nodes
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
    with pytest.raises(RuntimeError):
        sharding.open_exchange(0, 8, lambda: None, "/ecc_hip_test_multinode_%d" % os.getpid())
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    with pytest.raises(E.EccError):
        sharding.ScalarExchange(3, 2, "/x")          # rank outside the world
    with pytest.raises(E.EccError):
        sharding.ScalarExchange(0, 1, "no_slash")    # shm names start with '/'
