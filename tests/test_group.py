"""Single-process multi-GPU group (ecc_group_* of the C ABI; BASELINE config 4's partitioning).

CPU: the shard arithmetic.  GPU (one device on the test box): a 1-rank group equals ecc_metric_evaluate_all bit for
bit; 2 and 3 ranks on the SAME device (own stream + host thread each) exercise the threaded multi-rank path: rank-order
sum of the shard sums, cost image assembled from the shards, data-parallel Radon intermediates, errors from a worker
thread reaching the caller."""
import numpy as np
import pytest


def test_pair_shard_arithmetic():
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import sharding
    for n_pairs in (0, 1, 7, 2016, 79800, 79801):
        for world in (1, 2, 3, 8, 64):
            nxt = 0
            for r in range(world):
                first, count = E.pair_shard(n_pairs, world, r)
                assert (first, count) == sharding.pair_range(r, world, n_pairs)
                assert first == nxt and count >= 0
                nxt = first + count
            assert nxt == n_pairs
            counts = [E.pair_shard(n_pairs, world, r)[1] for r in range(world)]
            assert max(counts) - min(counts) <= 1
    assert E.pair_shard(79800, 8, 3) == (29925, 9975)  # BASELINE config 4: 9 975 pairs per GPU


def test_balanced_shards_model():
    """ecc_pair_shards_balanced against a numpy statement of its cost model (weights 1 + 2.8 kappa_max, 5.2 for
    kappa_max > pi/4; kappa_max from the source positions as in computeK01): contiguous, complete, every shard within
    one pair's weight of the mean cost, and -- for a circular scan -- fewer pairs for the first ranks."""
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import synthetic
    n, S = 400, 1024
    Ps = synthetic.short_scan(n, S, S, 0.308)
    radius = E.host_object_radius(Ps[0], S, S)
    C = np.stack([E.host_source_position(P).astype(np.float64) for P in Ps])
    iu = np.triu_indices(n, 1)
    a, b = C[iu[0]], C[iu[1]]

    def w2(p, q):
        return a[:, p] * b[:, q] - a[:, q] * b[:, p]
    s2 = np.sqrt(w2(1, 2) ** 2 + w2(0, 2) ** 2 + w2(0, 1) ** 2)
    s3 = np.sqrt(w2(0, 3) ** 2 + w2(1, 3) ** 2 + w2(2, 3) ** 2)
    dist = s2 / s3
    kmax = np.where(dist > radius, np.arcsin(np.minimum(radius / dist, 1.0)), np.pi / 2)
    w = np.where(kmax > np.pi / 4, 5.2, 1.0 + 2.8 * kmax)
    n_pairs = n * (n - 1) // 2
    for world in (1, 2, 4, 8, 7):
        bnd = E.pair_shards_balanced(Ps, radius, world)
        assert bnd[0] == 0 and bnd[-1] == n_pairs and len(bnd) == world + 1 and all(x <= y for x, y in zip(bnd, bnd[1:]))
        cost = np.array([w[bnd[r]:bnd[r + 1]].sum() for r in range(world)])
        assert np.all(np.abs(cost - w.sum() / world) <= 5.3), (world, cost)
    bnd = E.pair_shards_balanced(Ps, radius, 8)
    counts = np.diff(bnd)
    assert counts[0] < counts[3] < counts[7] and counts[0] < 0.8 * n_pairs / 8  # the expensive pairs come first
    with pytest.raises(E.EccError):
        E.pair_shards_balanced(Ps[:1], radius, 2)


@pytest.mark.gpu
def test_one_rank_group_is_the_plain_metric(gpu_ctx, small_scan):
    import epipolarconsistency_amd as E
    s = small_scan
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in s["dtrs"]]
    m = E.MetricRadonIntermediate(gpu_ctx, s["Ps"], dtrs)
    g = E.Group([0])
    assert len(g) == 1
    gm = E.GroupMetricRadonIntermediate(g, s["Ps"], dtrs)
    cost_a, cost_b = np.full((8, 8), 3.0, np.float32), np.full((8, 8), 3.0, np.float32)
    a, b = m.evaluate(cost_a), gm.evaluate(cost_b)
    assert a == b and np.array_equal(cost_a, cost_b)
    assert gm.getObjectRadius() == m.getObjectRadius()
    moved = list(s["Ps"])
    moved[3] = moved[3] @ E.geometry.rigid_transform(tx=1.0, ry=0.02)
    assert gm.setProjectionMatrices(moved).evaluate() == m.setProjectionMatrices(moved).evaluate() != a
    gm.close()
    g.close()
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3])
def test_multi_rank_group_on_one_device(gpu_ctx, oracle_mod, small_scan, ranks):
    import epipolarconsistency_amd as E
    s = small_scan
    g = E.Group([0] * ranks)
    # data-parallel Radon intermediates: bit-exact like the single-context path
    dtrs = g.compute_batch(s["imgs"], s["n_alpha"], s["n_t"])
    for k in (0, 2, 5, 7):
        assert np.array_equal(dtrs[k].readback(), s["dtrs"][k])
    gm = E.GroupMetricRadonIntermediate(g, s["Ps"], dtrs)
    m = E.MetricRadonIntermediate(gpu_ctx, s["Ps"], [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"])
                                                     for d in s["dtrs"]])
    total, vals = m.evaluate_range(0, 28, want_pairs=True)
    bnd = m.balanced_shards(ranks)  # the group cuts the pair range into cost-balanced contiguous shards
    assert bnd[0] == 0 and bnd[-1] == 28 and bnd == E.pair_shards_balanced(s["Ps"], m.getObjectRadius(), ranks)
    parts = [m.evaluate_range(bnd[r], bnd[r + 1] - bnd[r]) for r in range(ranks)]
    cost = np.full((8, 8), -1.0, np.float32)
    mean = gm.evaluate(cost)
    acc = 0.0
    for p in parts:
        acc += p
    assert mean == acc / 28  # rank-order float64 sum of the shard sums
    assert abs(mean - total / 28) <= 1e-13 * mean
    iu = np.triu_indices(8, 1)
    assert np.array_equal(cost[iu[1], iu[0]], vals) and np.all(cost[iu] == -1.0) and np.all(np.diag(cost) == -1.0)
    want = oracle_mod.evaluate_all(s["Ps"], s["dtrs"], s["n_u"], s["n_v"])
    assert abs(mean - want["mean"]) <= 1e-5 * want["mean"]
    # repeated evaluations (the worker threads go to sleep in between and are woken again), same bits
    import time
    for pause in (0.0, 0.0, 0.05):
        time.sleep(pause)
        assert gm.evaluate() == mean
    # parameters and sampling mode reach every rank
    gm.setSampling("reference")
    assert abs(gm.evaluate() - want["mean"]) <= 1e-7 * want["mean"]
    gm.setSampling("polynomial").setObjectRadius(25.0, 0.004)
    want2 = oracle_mod.evaluate_all(s["Ps"], s["dtrs"], s["n_u"], s["n_v"], object_radius_mm=25.0, dkappa=0.004)
    assert abs(gm.evaluate() - want2["mean"]) <= 1e-5 * want2["mean"]
    # an error raised on a worker thread reaches the caller with its message
    with pytest.raises(E.EccError) as ei:
        gm.setProjectionMatrices(s["Ps"] + s["Ps"]).evaluate()  # 16 matrices, 8 dtrs
    assert "fewer Radon intermediates" in str(ei.value)
    gm.setProjectionMatrices(s["Ps"])
    gm.setObjectRadius(0.0, 0.0)
    assert gm.evaluate() == mean
    assert abs(gm.rebalance().evaluate() - mean) <= 1e-13 * mean  # same geometry: the same boundaries again
    # independent evaluations of several poses, pose p on rank p mod G: each bit-identical to the single-device value,
    # and the group's own matrices are still current afterwards
    poses, want_p = [], []
    for k in range(5):
        Pk = list(s["Ps"])
        Pk[k + 1] = Pk[k + 1] @ E.geometry.rigid_transform(tx=0.5 * k, rz=0.01 * k)
        poses.append(Pk)
        want_p.append(m.setSampling("polynomial").setProjectionMatrices(Pk).evaluate())
    got_p = gm.evaluate_poses(poses)
    assert list(got_p) == want_p and abs(got_p[0] - mean) <= 1e-13 * mean
    assert gm.evaluate() == mean
    gm.close()
    m.close()
    del dtrs
    g.close()


@pytest.mark.gpu
def test_replica_path_on_one_device(gpu_ctx, small_scan):
    """ecc_group_debug_force_replica(1): every rank copies the Radon-intermediate stack although it is already on its device --
    the code the ranks of a real multi-GPU group run (allocation, device-to-device copies, read-back probes against the
    source, metrics built on the copy), minus the peer-to-peer flavour of the copy."""
    import epipolarconsistency_amd as E
    s = small_scan
    from epipolarconsistency_amd import _lib
    _lib.lib().ecc_group_debug_force_replica(1)
    try:
        g = E.Group([0, 0])
        dtrs = g.compute_batch(s["imgs"], s["n_alpha"], s["n_t"])
        gm = E.GroupMetricRadonIntermediate(g, s["Ps"], dtrs)
    finally:
        _lib.lib().ecc_group_debug_force_replica(0)
    m = E.MetricRadonIntermediate(gpu_ctx, s["Ps"], [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"])
                                                     for d in s["dtrs"]])
    bnd = m.balanced_shards(2)
    parts = [m.evaluate_range(bnd[r], bnd[r + 1] - bnd[r]) for r in range(2)]
    assert gm.evaluate() == (parts[0] + parts[1]) / 28
    moved = list(s["Ps"])
    moved[5] = moved[5] @ E.geometry.rigid_transform(ty=0.8, rx=0.01)
    m.setProjectionMatrices(moved)
    parts = [m.evaluate_range(bnd[r], bnd[r + 1] - bnd[r]) for r in range(2)]
    assert gm.setProjectionMatrices(moved).evaluate() == (parts[0] + parts[1]) / 28
    gm.close()
    g.close()
    m.close()
