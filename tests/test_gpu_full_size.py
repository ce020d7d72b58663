"""BASELINE.json's full configuration (400 views, 1024x1024, 768x768 bins, 79 800 pairs, N_kappa = 1448) through
the C ABI: size-independent properties of the metric, plus a bounded oracle check on a random sample of pairs.
The whole-problem comparison against the oracle is in test_gpu_parity.py (every 4th view) and in bench.py (all pairs,
`parity_rel_err_vs_oracle_on_sample`)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, S, B = 400, 1024, 768


@pytest.fixture(scope="module")
def full_scan(gpu_ctx):
    import torch
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import synthetic
    Ps = synthetic.short_scan(N, S, S, 0.308)
    dev = torch.device("cuda", 0)
    slabs = torch.zeros((N, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
    phantom = synthetic.sphere_phantom()
    for a in range(0, N, 50):
        imgs = synthetic.projections_torch(Ps[a:a + 50], S, S, phantom, dev)
        torch.cuda.synchronize()
        keep = E.RadonIntermediate.compute_into(gpu_ctx, imgs, slabs[a:a + 50], B, B)
        gpu_ctx.synchronize()
        del keep, imgs
    dtrs = [E.RadonIntermediate.wrap_device(gpu_ctx, slabs[k], B, B, S, S) for k in range(N)]
    m = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs)
    yield dict(Ps=Ps, dtrs=dtrs, metric=m, slabs=slabs)
    m.close()


def test_shards_cost_image_and_determinism(full_scan):
    """Sum of 8 range shards == all-pairs sum; the cost image holds the pair values at index i + j*n; a second
    evaluation returns the same bits; perturbing one view and restoring it restores the value exactly."""
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import geometry, sharding
    m, Ps = full_scan["metric"], full_scan["Ps"]
    n_pairs = N * (N - 1) // 2
    cost = np.full((N, N), -1.0, np.float32)
    mean = m.evaluate(cost)
    assert mean > 0 and np.isfinite(mean)
    assert m.evaluate() == mean
    total, vals = m.evaluate_range(0, n_pairs, want_pairs=True)
    assert total / n_pairs == mean
    parts = [m.evaluate_range(*sharding.pair_range(r, 8, n_pairs)) for r in range(8)]
    assert abs(sum(parts) - total) <= 1e-12 * total
    # small launches run the pair-geometry kernel in its 8-lanes-per-fit form (k01_kernel<8>, at most 4096 pairs): every
    # pair value is the one of the 79 800-pair launch, bit for bit
    for first, count in ((0, 4096), (20000, 3000), (75704, 4096), (399, 399), (12345, 7)):
        t, v = m.evaluate_range(first, count, want_pairs=True)
        assert np.array_equal(v, vals[first:first + count]), (first, count)
    # cost image: entry [j, i] for i < j, everything else untouched (ref: ...RadonIntermediate.cu:250,269)
    iu = np.triu_indices(N, 1)  # (i, j) with i < j in get_ij order
    assert np.array_equal(cost[iu[1], iu[0]], vals)
    assert np.all(cost[iu] == -1.0) and np.all(np.diag(cost) == -1.0)
    assert E.get_ij(n_pairs - 1, N) == (N - 2, N - 1)
    # one view moved: the value changes; moved back: the same bits again
    moved = list(Ps)
    moved[200] = Ps[200] @ geometry.rigid_transform(tx=2.0, rz=0.01)
    m.setProjectionMatrices(moved)
    assert abs(m.evaluate() - mean) > 1e-6 * mean
    m.setProjectionMatrices(Ps)
    assert m.evaluate() == mean


def test_random_pairs_against_oracle(full_scan, oracle_mod):
    """300 random pairs at full size: index-list evaluation vs the oracle on read-back dtrs (mean within 1e-5,
    single pairs within the fp32 noise floor of this size), and the metric's symmetry under swapping the views."""
    m, Ps, dtrs = full_scan["metric"], full_scan["Ps"], full_scan["dtrs"]
    rng = np.random.default_rng(11)
    pairs = np.array([sorted(rng.choice(N, 2, replace=False)) for _ in range(300)], np.int32)
    idx = np.stack([pairs[:, 0], pairs[:, 1], pairs[:, 0], pairs[:, 1]], 1).astype(np.int32)
    out = np.empty(len(idx), np.float32)
    mean = m.evaluate(idx, out)
    used = sorted(set(pairs.ravel().tolist()))
    host = {v: dtrs[v].readback() for v in used}
    remap = {v: k for k, v in enumerate(used)}
    idx_o = np.array([[remap[a], remap[b], remap[a], remap[b]] for a, b in pairs], np.int32)
    want = oracle_mod.evaluate_pairs([Ps[v] for v in used], [host[v] for v in used], S, S, idx_o)
    assert abs(mean - want["mean"]) <= 1e-5 * want["mean"], (mean, want["mean"])
    np.testing.assert_allclose(out, want["pairs"], rtol=2e-3)
    # swapped views: the same pair with the roles of view 0 and view 1 exchanged
    swapped = idx[:, [1, 0, 3, 2]].copy()
    out_s = np.empty(len(idx), np.float32)
    mean_s = m.evaluate(swapped, out_s)
    assert abs(mean_s - mean) <= 1e-5 * mean
    np.testing.assert_allclose(out_s, out, rtol=2e-3)
    # subset semantics: all pairs among a set of views == the cost-image entries of those pairs
    views = sorted(int(v) for v in rng.choice(N, 12, replace=False))
    cost = np.zeros((N, N), np.float32)
    m.evaluate(cost)
    entries = [cost[b, a] for k, a in enumerate(views) for b in views[k + 1:]]
    assert abs(m.evaluate(set(views)) - float(np.mean(np.asarray(entries, np.float64)))) <= 2e-6 * np.mean(entries)


def test_multi_workgroup_sum_is_stable_under_repetition(full_scan):
    """The final float64 sum runs over 16 workgroups of one launch with a last-arriver combine (pairs_kernel.hip,
    sum_pairs_split_kernel).  400 evaluations alternating between two poses: every result for a pose must carry the
    same bits (a stale partial from the other pose or the previous launch would move the sum by percents), and agree
    with the float64 sum of the pair values."""
    from epipolarconsistency_amd import geometry
    m, Ps = full_scan["metric"], full_scan["Ps"]
    n_pairs = N * (N - 1) // 2
    moved = list(Ps)
    moved[123] = Ps[123] @ geometry.rigid_transform(ty=1.5, rx=0.005)
    seen = {0: set(), 1: set()}
    for k in range(400):
        m.setProjectionMatrices(moved if k & 1 else Ps)
        seen[k & 1].add(m.evaluate())
    assert len(seen[0]) == 1 and len(seen[1]) == 1 and seen[0] != seen[1]
    m.setProjectionMatrices(Ps)
    total, vals = m.evaluate_range(0, n_pairs, want_pairs=True)
    assert abs(total - float(np.sum(vals.astype(np.float64)))) <= 1e-13 * total
    assert total / n_pairs == next(iter(seen[0]))


def test_pose_batch_at_full_size(full_scan, oracle_mod):
    """BASELINE config 5's shape at full size (ref: Gui/Visualization.h:59-112): poses of view 200 (and a few of two views) as ONE
    batched record / pair / sum launch each -- the sixteen-slice sum over 79 800 values with 399 or 797 substituted -- against
    setProjectionMatrices + evaluate per pose, bit for bit; one pose against the oracle on the read-back Radon intermediates."""
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import geometry
    m, Ps = full_scan["metric"], full_scan["Ps"]
    P0 = E.pack_projection_matrices(Ps)
    m.setProjectionMatrices(P0)
    K = 37  # not a divisor of anything
    views, rows, poses = [], [], []
    for k in range(K):
        vk = [200] if k % 5 else [3 + k, 200]
        P = P0.copy()
        for v in vk:
            P[v] = (Ps[v] @ geometry.rigid_transform(tx=-2.0 + 0.1 * k, ry=0.0007 * k, rz=0.002 * (v % 3))).T.reshape(12)
        views.append(vk)
        rows.append(P[vk].copy())
        poses.append(P)
    got = m.evaluate_pose_deltas(views, rows)
    assert m.last_batched_poses() == K
    base = m.evaluate()
    m.setPoseBatching(False)
    want = np.array([m.setProjectionMatrices(P).evaluate() for P in poses])
    m.setPoseBatching(True)
    assert np.array_equal(got, want), np.flatnonzero(got != want)
    assert len(set(want.tolist())) == K
    # the full-matrix form, rank 1 of 3, starting from the last pose's matrices (most poses are deltas of the FIRST pose then)
    part = m.evaluate_poses(np.ascontiguousarray(np.stack(poses)), first=1, stride=3)
    assert np.array_equal(part[1::3], want[1::3]) and not part[0::3].any() and not part[2::3].any()
    m.setProjectionMatrices(P0)
    assert m.evaluate() == base
    # one pose against the oracle (every 8th view: 1 225 pairs among them, the moved view included)
    sub = list(range(0, N, 8))
    assert 200 in sub
    Pk = [poses[7][v].reshape(4, 3).T.copy() for v in sub]
    host = [full_scan["dtrs"][v].readback() for v in sub]
    ref = oracle_mod.evaluate_all(Pk, host, S, S)["mean"]
    m.setProjectionMatrices(poses[7])
    assert abs(m.evaluate(set(sub)) - ref) <= 1e-5 * ref
    m.setProjectionMatrices(P0)


def test_device_bytes_of_the_defaults(full_scan):
    """ecc_metric_device_bytes: what a 400-view metric holds beside the 0.99-GB stack -- the numbers
    INTEGRATION.md and include/ecc_hip.h state: row-paired copies 2x, row-quad copies 4x the slab."""
    b = full_scan["metric"].device_bytes()
    pitch = (B + 2 + 31) // 32 * 32
    assert b["paired_copies"] == N * (B + 1) * pitch * 8 == 1968640000
    assert b["quad_copies"] in (0, N * ((B + 1 + 3) // 4) * pitch * 64)   # built while they fit a quarter of the free memory
    assert 0 < b["other"] < 1 << 30
