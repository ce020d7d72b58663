"""ecc_metric_evaluate_pose_deltas / ecc_metric_evaluate_poses[_strided] (csrc/ecc_poses.hip): K poses of one data set as ONE
e1 launch, ONE record launch, ONE pair launch and ONE segmented float64 sum (BASELINE config 5; ref: Gui/Visualization.h:59-112
plotCostFunction -- 100 steps x 6 parameters -- through Gui/SingleImageMotion.h:84-90, which the reference evaluates one
setProjectionMatrices + evaluate at a time).

The contract: every mean is BIT-IDENTICAL to ecc_metric_set_projections + ecc_metric_evaluate_all on that pose's matrices --
whatever the number of views (both forms of the all-pairs sum: one workgroup below 32 768 pairs, sixteen slices from there on;
counts with and without a tail past the last float4), the sampling mode, the number of moved views per pose (one, several
including pairs of two moved views, none, more than the batch takes), a moved view 0 under the automatic object radius,
batches that split (ECC_POSE_BATCH_MAX_ENTRIES), strides (rank r of N), and whatever the metric did before."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scan(gpu_ctx, n, S=128, B=48, seed=5):
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import synthetic
    rng = np.random.default_rng(seed)
    Ps = synthetic.short_scan(n, S, S, 0.308 * 1024 / S)
    base = [E.RadonIntermediate.from_host(gpu_ctx, rng.standard_normal((B, B)).astype(np.float32), S, S) for _ in range(7)]
    return Ps, base, [base[v % 7] for v in range(n)]


def _perturb(P12, k, v):
    import epipolarconsistency_amd as E
    T = E.geometry.rigid_transform(tx=0.11 * (k + 1), ty=-0.05 * (v % 5), rz=0.0015 * (k + 1), rx=0.0007 * (v % 3))
    return (P12.reshape(4, 3).T @ T).T.reshape(12)


def _poses(P0, n, K, moved_of):
    """K full pose arrays (n, 12) + their sparse form; moved_of(k) -> ascending list of views pose k moves."""
    poses, views, rows = [], [], []
    for k in range(K):
        P = P0.copy()
        vk = sorted(set(moved_of(k)))
        for v in vk:
            P[v] = _perturb(P0[v], k, v)
        poses.append(P)
        views.append(vk)
        rows.append(P[vk].copy() if vk else np.zeros((0, 12)))
    return poses, views, rows


@pytest.mark.parametrize("n,mode", [(2, "polynomial"), (3, "auto"), (9, "auto"), (9, "polynomial"), (34, "auto"), (34, "per_sample"),
                                    (67, "polynomial"), (130, "auto"), (257, "polynomial"), (258, "auto")])
def test_deltas_have_the_sequential_bits(gpu_ctx, n, mode):
    """n = 257 / 258: 32 896 / 33 153 pairs (the sixteen-slice sum, without and with a tail); 130: 8 385 pairs (one workgroup, two
    staged chunks, tail of 1); 67: 2 211 (tail 3); 34: 561; 9: 36; 3: 3 (tail only); 2: 1."""
    import epipolarconsistency_amd as E
    Ps, base, dtrs = _scan(gpu_ctx, n, B=32 if n > 100 else 48)
    P0 = E.pack_projection_matrices(Ps)
    K = 23 if n < 200 else 11

    def moved_of(k):
        if k % 7 == 3:
            return []                                   # the base itself
        if k % 5 == 0 and n > 4:
            return [k % n, (3 * k + 1) % n, (n - 1 - k) % n, n // 2]   # several, pairs of two moved views among them
        if k % 11 == 6:
            return [n - 1]
        return [(2 * k + 1) % n]
    poses, views, rows = _poses(P0, n, K, moved_of)
    a = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setSampling(mode)
    b = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setSampling(mode).setPoseBatching(False)
    want = np.array([b.setProjectionMatrices(P).evaluate() for P in poses])
    got = a.evaluate_pose_deltas(views, rows)
    assert np.array_equal(got, want), (n, mode, np.flatnonzero(got != want), (got - want)[got != want])
    # (a pose that moves view 0 changes the automatic object radius: evaluated the sequential way inside the call)
    assert K - sum(1 for vk in views if 0 in vk) <= a.last_batched_poses() <= K
    assert a.evaluate() == b.setProjectionMatrices(P0).evaluate()  # the current matrices are still the base
    # again from kept state (the base's values are cached), then from a different base
    assert np.array_equal(a.evaluate_pose_deltas(views[::-1], rows[::-1]), want[::-1])
    P1 = poses[5 % K]
    a.setProjectionMatrices(P1)
    poses2, views2, rows2 = _poses(P1, n, 6, lambda k: [(5 * k + 2) % n])
    want2 = np.array([b.setProjectionMatrices(P).evaluate() for P in poses2])
    assert np.array_equal(a.evaluate_pose_deltas(views2, rows2), want2)
    a.close(); b.close()
    for d in base:
        d.close()


def test_dense_poses_strides_and_leftovers(gpu_ctx):
    """ecc_metric_evaluate_poses_strided: the library finds the moved views itself; K that does not divide over the ranks; poses
    with more moved views than the batch takes (40 > 32) and a pose that is a different trajectory altogether go the sequential
    way inside the same call; the last evaluated pose's matrices stay current."""
    import epipolarconsistency_amd as E
    n = 60
    Ps, base, dtrs = _scan(gpu_ctx, n)
    P0 = E.pack_projection_matrices(Ps)

    def moved_of(k):
        if k == 4:
            return list(range(40))          # more than ECC_POSE_BATCH_MAX_MOVED
        if k == 9:
            return list(range(n))           # every view
        if k % 3 == 0:
            return [k % n, (k + 7) % n]     # two moved views
        return [(7 * k + 3) % n]
    poses, _, _ = _poses(P0, n, 29, moved_of)
    ref = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setPoseBatching(False)
    want = np.array([ref.setProjectionMatrices(P).evaluate() for P in poses])
    a = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs)
    got = a.evaluate_poses(poses)
    assert np.array_equal(got, want), np.flatnonzero(got != want)
    assert a.last_batched_poses() == 26  # 29 - the two above - pose 0, which moves view 0 (automatic object radius)
    assert a.evaluate() == ref.setProjectionMatrices(poses[-1]).evaluate()
    for world in (2, 3, 4):  # rank r of N: poses r, r + N, ...
        total = np.zeros(len(poses))
        for r in range(world):
            m = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs)
            part = m.evaluate_poses(poses, first=r, stride=world)
            assert np.all(part[[q for q in range(len(poses)) if q % world != r]] == 0.0)
            total += part
            m.close()
        assert np.array_equal(total, want), world
    # a metric that has never seen these matrices: the first pose becomes the base
    Pfar = [P @ E.geometry.rigid_transform(tz=3.0, ry=0.01) for P in Ps]
    far0 = E.pack_projection_matrices(Pfar)
    poses_far, _, _ = _poses(far0, n, 8, lambda k: [(k + 1) % n] if k else [])
    c = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs)
    want_far = np.array([ref.setProjectionMatrices(P).evaluate() for P in poses_far])
    assert np.array_equal(c.evaluate_poses(poses_far), want_far)
    assert c.last_batched_poses() == 8
    # batching off: the two-deep launches of rounds 4-5
    c.setPoseBatching(False)
    assert np.array_equal(c.evaluate_poses(poses), want) and c.last_batched_poses() == 0
    a.close(); c.close(); ref.close()
    for d in base:
        d.close()


def test_moved_view_zero_and_automatic_radius(gpu_ctx):
    """The automatic object radius is a function of the FIRST matrix (ref: EpipolarConsistency.cpp:76-84): a pose that moves
    view 0 changes every pair's record unless the radius happens to stay -- such a pose is evaluated the sequential way; with a
    user radius view 0 is a moved view like any other."""
    import epipolarconsistency_amd as E
    n = 24
    Ps, base, dtrs = _scan(gpu_ctx, n)
    P0 = E.pack_projection_matrices(Ps)
    poses, views, rows = _poses(P0, n, 9, lambda k: [0] if k % 2 == 0 else [0, 5 + k])
    for radius in (None, 80.0):
        a = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs)
        b = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setPoseBatching(False)
        if radius:
            a.setObjectRadius(radius)
            b.setObjectRadius(radius)
        want = np.array([b.setProjectionMatrices(P).evaluate() for P in poses])
        got = a.evaluate_pose_deltas(views, rows)
        assert np.array_equal(got, want), (radius, got - want)
        assert a.last_batched_poses() == 9 if radius else a.last_batched_poses() < 9  # (a pose whose radius rounds to the base's float is a delta)
        assert np.array_equal(a.evaluate_poses(poses), want)
        assert a.evaluate() == b.evaluate()
        a.close(); b.close()
    for d in base:
        d.close()


def test_modes_and_interleaving(gpu_ctx):
    """use_corr, a user dkappa, the non-default modes, the pose-delta mode and record reuse off; batches between ordinary
    evaluations, index lists and in-place changes of the Radon intermediates (the base's kept values must not go stale)."""
    import epipolarconsistency_amd as E
    n = 40
    Ps, base, dtrs = _scan(gpu_ctx, n)
    P0 = E.pack_projection_matrices(Ps)
    poses, views, rows = _poses(P0, n, 14, lambda k: [(3 * k + 1) % n] if k % 4 else [k % n, (k + 11) % n])
    setups = [lambda m: m.useCorrelation(True), lambda m: m.setEpipolarPlaneStep(0.004), lambda m: m.setSampling("reference"),
              lambda m: m.setIncremental(True), lambda m: m.setRecordReuse(False), lambda m: m.setSmallEval(False)]
    for setup in setups:
        a = setup(E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs))
        b = setup(E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs)).setPoseBatching(False)
        want = np.array([b.setProjectionMatrices(P).evaluate() for P in poses])
        assert np.array_equal(a.evaluate_pose_deltas(views, rows), want)
        a.close(); b.close()
    a = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs)
    b = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setPoseBatching(False)
    want = np.array([b.setProjectionMatrices(P).evaluate() for P in poses])
    idx = np.array([[1, 7, 1, 7], [3, 30, 3, 30], [12, 13, 12, 13]], np.int32)
    for step in range(3):
        assert np.array_equal(a.evaluate_pose_deltas(views, rows), want)
        assert a.evaluate() == b.setProjectionMatrices(P0).evaluate()
        assert a.evaluate(idx) == b.evaluate(idx)
        assert a.setProjectionMatrices(poses[step]).evaluate() == want[step]
        a.setProjectionMatrices(P0)
    # parameters change between batches
    a.setObjectRadius(70.0); b.setObjectRadius(70.0)
    want_r = np.array([b.setProjectionMatrices(P).evaluate() for P in poses])
    assert np.array_equal(a.evaluate_pose_deltas(views, rows), want_r) and not np.array_equal(want_r, want)
    a.close(); b.close()
    for d in base:
        d.close()


def test_argument_checks(gpu_ctx):
    import epipolarconsistency_amd as E
    n = 6
    Ps, base, dtrs = _scan(gpu_ctx, n)
    P0 = E.pack_projection_matrices(Ps)
    a = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs)
    with pytest.raises(E.EccError):
        a.evaluate_pose_deltas([[2, 2]], [np.stack([P0[2], P0[2]])])     # not strictly ascending
    with pytest.raises(E.EccError):
        a.evaluate_pose_deltas([[n]], [P0[:1]])                          # outside [0, n)
    with pytest.raises(E.EccError):
        a.evaluate_pose_deltas([[3, 1]], [P0[:2]])
    assert len(a.evaluate_pose_deltas([], [])) == 0
    assert a.evaluate_pose_deltas([[]], [np.zeros((0, 12))])[0] == a.evaluate()
    a.close()
    for d in base:
        d.close()


def test_deltas_with_batching_off_and_large_moved_sets(gpu_ctx):
    """ecc_metric_evaluate_pose_deltas is a complete entry point whatever the batch takes: with ecc_metric_set_pose_batching(0),
    and for poses that move more views than the batch handles (33 and 40 > 32), the poses are expanded and evaluated the
    sequential way inside the call -- the same values, and the current matrices are the base again afterwards."""
    import epipolarconsistency_amd as E
    n = 50
    Ps, base, dtrs = _scan(gpu_ctx, n)
    P0 = E.pack_projection_matrices(Ps)
    poses, views, rows = _poses(P0, n, 7, lambda k: [list(range(1, 34)), [4], list(range(5, 45)), [], [9, 30], [49], list(range(0, 50, 2))][k])
    ref = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setPoseBatching(False)
    want = np.array([ref.setProjectionMatrices(P).evaluate() for P in poses])
    base_value = ref.setProjectionMatrices(P0).evaluate()
    a = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs)
    assert np.array_equal(a.evaluate_pose_deltas(views, rows), want)
    # 33 and 40 moved views: sequential; so is the last pose (25 views, view 0 among them: the automatic object radius changes)
    assert a.last_batched_poses() == 4 and a.evaluate() == base_value
    a.setPoseBatching(False)
    assert np.array_equal(a.evaluate_pose_deltas(views, rows), want)
    assert a.last_batched_poses() == 0 and a.evaluate() == base_value
    a.close(); ref.close()
    for d in base:
        d.close()


@pytest.mark.parametrize("n", [520, 1030])
def test_many_views_stage_a_slice_in_several_chunks(gpu_ctx, n):
    """134 940 / 529 935 pairs: a slice of the sixteen-slice sum is 8 434 / 33 121 floats -- more than the 8 192 the segmented sum
    stages at a time (two / five chunks per slice; a thread's own order k, k + 1024, ... runs across the chunk boundaries), and more
    than 512 views: launch_range's skip mask no longer applies, the base evaluation refits on one stream."""
    import epipolarconsistency_amd as E
    Ps, base, dtrs = _scan(gpu_ctx, n, S=96, B=32)
    P0 = E.pack_projection_matrices(Ps)
    poses, views, rows = _poses(P0, n, 7, lambda k: [[n - 1], [1, n // 2], [], [n // 3], [7, 8, 9, n - 2], [n // 2], [2]][k])
    a = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs)
    b = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setPoseBatching(False)
    want = np.array([b.setProjectionMatrices(P).evaluate() for P in poses])
    got = a.evaluate_pose_deltas(views, rows)
    assert np.array_equal(got, want), (n, got - want)
    assert a.last_batched_poses() == 7 and len(set(want.tolist())) == 7
    assert np.array_equal(a.evaluate_poses(poses, first=1, stride=2)[1::2], want[1::2])
    a.close(); b.close()
    for d in base:
        d.close()


def test_more_entries_than_one_batch_takes(gpu_ctx):
    """3 600 poses of 300 views are 1 080 000 grid entries: more than ECC_POSE_BATCH_MAX_ENTRIES (2^20) -- the call runs them as two
    batches (3 495 + 105 columns) over the same base values.  Reference: the pose-delta mode, one pose at a time (bit-identical to
    full evaluations: tests/test_gpu_incremental.py)."""
    import epipolarconsistency_amd as E
    n, K = 300, 3600
    Ps, base, dtrs = _scan(gpu_ctx, n, S=96, B=32)
    P0 = E.pack_projection_matrices(Ps)
    rng = np.random.default_rng(2)
    moved = rng.integers(1, n, size=K)
    rows = np.stack([_perturb(P0[v], k % 97, int(v)) for k, v in enumerate(moved)])
    a = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs)
    got = a.evaluate_pose_deltas_packed(np.arange(K + 1, dtype=np.int32), moved.astype(np.int32), rows)
    assert a.last_batched_poses() == K
    b = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setPoseBatching(False).setIncremental(True)
    b.evaluate()
    P = P0.copy()
    want = np.empty(K)
    for k in range(K):
        P[moved[k]] = rows[k]
        want[k] = b.setProjectionMatrices(P).evaluate()
        P[moved[k]] = P0[moved[k]]
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]
    a.close(); b.close()
    for d in base:
        d.close()
