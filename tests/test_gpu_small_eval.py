"""ecc_metric_set_small_eval (default on): evaluations of at most 4096 pairs run as ONE launch (small_eval_kernel.hip) instead
of the stream-ordered E1 / K01 / pairs / sum launches -- what the reference does with two kernels, two device-wide syncs and
a host loop (ref: EpipolarConsistencyRadonIntermediate.cu:300-409, ...RadonIntermediate.cpp:197-224), for callers that
evaluate a handful of pairs per objective call (ref: tools/FluoroTracking/FluoroTracking.cpp:179-211).
The contract tested here: every mean, every pair value, every cost image is BIT-IDENTICAL to the multi-launch path
(setSmallEval(False)), in every sampling mode, for all-pairs evaluations, ranges, index lists and their mixtures."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _moved(Ps, views, k=1.0):
    import epipolarconsistency_amd as E
    out = list(Ps)
    for v in views:
        out[v] = out[v] @ E.geometry.rigid_transform(tx=0.7 * k, ty=-0.3 * k, rz=0.01 * k, ry=0.004 * (v + 1))
    return out


def _scan(gpu_ctx, n, S=128, B=64, seed=3, filt=None):
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import synthetic
    rng = np.random.default_rng(seed)
    Ps = synthetic.short_scan(n, S, S, 0.308 * 1024 / S)
    base = [E.RadonIntermediate.from_host(gpu_ctx, rng.standard_normal((B, B)).astype(np.float32), S, S,
                                          **({} if filt is None else {"filter": filt})) for _ in range(5)]
    return Ps, base, [base[v % 5] for v in range(n)]


def _pair(gpu_ctx, Ps, dtrs, mode):
    import epipolarconsistency_amd as E
    on = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setSampling(mode)
    off = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setSampling(mode).setSmallEval(False)
    return on, off


@pytest.mark.parametrize("n", [2, 8, 33, 46, 64, 91])  # 1 / 28 / 528 / 1035 / 2016 / 4095 pairs: four, two, one wave per pair
@pytest.mark.parametrize("mode", ["auto", "polynomial", "per_sample", "reference"])
def test_all_pairs_bit_identical(gpu_ctx, n, mode):
    if mode == "reference" and n > 64:
        pytest.skip("reference arithmetic on > 2048 pairs: seconds per evaluation, covered at 2016")
    Ps, base, dtrs = _scan(gpu_ctx, n)
    on, off = _pair(gpu_ctx, Ps, dtrs, mode)
    n_pairs = n * (n - 1) // 2
    for step, views in enumerate(([], [n // 2], [0], list(range(0, n, 2)), [])):
        P1 = _moved(Ps, views, 0.3 * (step + 1))
        a = on.setProjectionMatrices(P1).evaluate()
        b = off.setProjectionMatrices(P1).evaluate()
        assert a == b, (n, mode, views, a, b)
        assert on.last_evaluated_pairs() == n_pairs
    ca, cb = np.full((n, n), 3.0, np.float32), np.full((n, n), 3.0, np.float32)
    assert on.evaluate(ca) == off.evaluate(cb)
    assert np.array_equal(ca, cb) and ca[0, 0] == 3.0 and ca[1, 0] != 3.0 and ca[0, 1] == 3.0
    sa, va = on.evaluate_range(0, n_pairs, want_pairs=True)
    sb, vb = off.evaluate_range(0, n_pairs, want_pairs=True)
    assert sa == sb and np.array_equal(va, vb)
    on.close(); off.close()
    for d in base:
        d.close()


def test_ranges_index_lists_and_parameters(gpu_ctx, small_scan, oracle_mod):
    import epipolarconsistency_amd as E
    s = small_scan
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in s["dtrs"]]
    want = oracle_mod.evaluate_all(s["Ps"], s["dtrs"], s["n_u"], s["n_v"])
    for mode in ("auto", "polynomial"):
        on, off = _pair(gpu_ctx, s["Ps"], dtrs, mode)
        assert abs(on.evaluate() - want["mean"]) <= 1e-5 * want["mean"]  # and it is the right number
        for first, count in ((0, 7), (7, 6), (13, 15), (27, 1), (3, 20)):
            a, va = on.evaluate_range(first, count, want_pairs=True)
            b, vb = off.evaluate_range(first, count, want_pairs=True)
            assert a == b and np.array_equal(va, vb), (mode, first, count)
        idx = np.array([[0, 1, 0, 1], [2, 5, 2, 5], [7, 3, 7, 3], [1, 6, 4, 2], [4, 5, 5, 4], [6, 0, 6, 0], [3, 3, 3, 3]], np.int32)
        oa, ob = np.empty(len(idx), np.float32), np.empty(len(idx), np.float32)
        assert on.evaluate(idx, oa) == off.evaluate(idx, ob) and np.array_equal(oa, ob)
        assert on.evaluate(idx) == off.evaluate(idx)  # without the values
        assert on.evaluate({1, 2, 4, 6, 7}) == off.evaluate({1, 2, 4, 6, 7})
        for change in (lambda m: m.setObjectRadius(25.0), lambda m: m.setdKappa(0.004), lambda m: m.setObjectRadius(0.0).setdKappa(0.0)):
            change(on); change(off)
            assert on.evaluate() == off.evaluate()
            assert on.evaluate(idx, oa) == off.evaluate(idx, ob) and np.array_equal(oa, ob)
        # use_corr is not fused: still the same numbers through the old path
        on.useCorrelation(True); off.useCorrelation(True)
        assert on.evaluate() == off.evaluate()
        on.close(); off.close()


def test_plain_line_integrals_and_many_samples(gpu_ctx):
    """Filter::None dtrs (no sign flip in the fold); a user-chosen dkappa with ~10 000 samples per pair does not fit the
    LDS stage and takes the old path -- same numbers either way."""
    import epipolarconsistency_amd as E
    Ps, base, dtrs = _scan(gpu_ctx, 12, filt=E.FILTER_NONE)
    for mode in ("auto", "polynomial", "per_sample"):
        on, off = _pair(gpu_ctx, Ps, dtrs, mode)
        assert on.evaluate() == off.evaluate()
        on.setdKappa(1e-4); off.setdKappa(1e-4)
        assert on.evaluate() == off.evaluate()
        on.close(); off.close()
    for d in base:
        d.close()


def test_index_lists_on_a_large_metric_and_interleaving_with_big_evaluations(gpu_ctx):
    """200 views: index lists of 1 / 199 / 512 / 3000 pairs go out as one launch while all-pairs evaluations (19 900 pairs)
    keep the stream-ordered path with its kept records; neither disturbs the other, whatever the order."""
    import epipolarconsistency_amd as E
    n = 200
    Ps, base, dtrs = _scan(gpu_ctx, n, B=48)
    rng = np.random.default_rng(5)
    lists = [np.array([(10, 150, 10, 150)], np.int32),
             np.array([(min(100, v), max(100, v), min(100, v), max(100, v)) for v in range(n) if v != 100], np.int32),
             np.array([(a, b, a, b) for a, b in (sorted(rng.choice(n, 2, replace=False)) for _ in range(512))], np.int32),
             np.array([(a, b, a, b) for a, b in (sorted(rng.choice(n, 2, replace=False)) for _ in range(3000))], np.int32)]
    for mode in ("auto", "polynomial"):
        on, off = _pair(gpu_ctx, Ps, dtrs, mode)
        for step in range(6):
            P1 = _moved(Ps, [(37 * step) % n] if step % 3 else list(range(step, n, 3)), 0.2 * (step + 1))
            on.setProjectionMatrices(P1); off.setProjectionMatrices(P1)
            for idx in lists if step % 2 else lists[::-1]:
                oa, ob = np.empty(len(idx), np.float32), np.empty(len(idx), np.float32)
                assert on.evaluate(idx, oa) == off.evaluate(idx, ob), (mode, step, len(idx))
                assert np.array_equal(oa, ob)
            assert on.evaluate() == off.evaluate()
            if step == 2:
                for x, y in zip(on.debug_geometry(), off.debug_geometry()):
                    assert np.array_equal(x, y)
        on.close(); off.close()
    for d in base:
        d.close()


def test_asynchronous_small_ranges(gpu_ctx, small_scan):
    import torch
    import epipolarconsistency_amd as E
    s = small_scan
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in s["dtrs"]]
    on, off = _pair(gpu_ctx, s["Ps"], dtrs, "polynomial")
    dev = torch.device("cuda", 0)
    sums = torch.zeros(16, dtype=torch.float64, device=dev)
    want = []
    for k in range(16):  # no synchronisation between the calls: the geometry buffer is guarded by an event
        P1 = _moved(s["Ps"], [k % 8] if k % 3 else [], 0.1 * (k + 1))
        on.setProjectionMatrices(P1)
        on.evaluate_range_async(k % 5, 28 - k % 5, sums[k:k + 1])
        want.append(off.setProjectionMatrices(P1).evaluate_range(k % 5, 28 - k % 5))
    gpu_ctx.synchronize()
    torch.cuda.synchronize()
    assert sums.cpu().tolist() == want
    assert on.evaluate() == off.evaluate()  # and a synchronous call afterwards
    on.close(); off.close()


def test_with_incremental_mode_and_group(gpu_ctx, small_scan):
    import epipolarconsistency_amd as E
    s = small_scan
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in s["dtrs"]]
    on, off = _pair(gpu_ctx, s["Ps"], dtrs, "auto")
    on.setIncremental(True); off.setIncremental(True)
    for k, views in enumerate(([], [3], [3], [0], [1, 2], [], [5])):
        P1 = _moved(s["Ps"], views, 0.25 * (k + 1))
        assert on.setProjectionMatrices(P1).evaluate() == off.setProjectionMatrices(P1).evaluate(), views
    on.close(); off.close()
    g = E.Group([0, 0, 0])
    gd = g.compute_batch(s["imgs"], s["n_alpha"], s["n_t"])
    gm = E.GroupMetricRadonIntermediate(g, s["Ps"], gd)
    ref = E.MetricRadonIntermediate(gpu_ctx, s["Ps"], dtrs).setSmallEval(False)
    bnd = ref.balanced_shards(3)
    for views in ([], [3], [0], [1, 2]):
        P1 = _moved(s["Ps"], views, 0.35)
        a = gm.setProjectionMatrices(P1).evaluate()
        ref.setProjectionMatrices(P1)
        parts = [ref.evaluate_range(bnd[r], bnd[r + 1] - bnd[r]) for r in range(3)]
        assert a == (parts[0] + parts[1] + parts[2]) / 28, views
    gm.close(); ref.close(); g.close()


def test_poses_two_deep_are_the_one_by_one_values(gpu_ctx):
    """ecc_metric_evaluate_poses: the same launches in the same order, only the host's waiting moves -- every mean bit-identical
    to setProjectionMatrices + evaluate per pose, at sizes that take the one-stream and the two-stream refit, with reuse
    off, and in the pose-delta mode (which evaluates them one at a time)."""
    import epipolarconsistency_amd as E
    for n, B in ((5, 48), (20, 48), (150, 32)):  # 10 / 190 pairs (one at a time: the one-launch path's sizes), 11 175 pairs (two-stream reuse)
        Ps, base, dtrs = _scan(gpu_ctx, n, B=B)
        P0 = E.pack_projection_matrices(Ps)
        poses = []
        for k in range(17):
            P = P0.copy()
            for v in ([k % n] if k % 4 else [k % n, (3 * k + 1) % n, 0]):
                P[v] = (P[v].reshape(4, 3).T @ E.geometry.rigid_transform(tx=0.1 * (k + 1), rz=0.002 * k)).T.reshape(12)
            poses.append(P)
        for setup in (lambda m: m, lambda m: m.setRecordReuse(False), lambda m: m.setIncremental(True), lambda m: m.setSampling("per_sample")):
            a = setup(E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setSampling("polynomial"))
            b = setup(E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setSampling("polynomial"))
            got = a.evaluate_poses(poses)
            want = np.array([b.setProjectionMatrices(P).evaluate() for P in poses])
            assert np.array_equal(got, want), (n, got - want)
            assert a.evaluate() == b.evaluate()  # the last pose's matrices stay current
            got2 = a.evaluate_poses(poses[::-1])  # and again, starting from kept state
            want2 = np.array([b.setProjectionMatrices(P).evaluate() for P in poses[::-1]])
            assert np.array_equal(got2, want2)
            a.close(); b.close()
        for d in base:
            d.close()


@pytest.mark.parametrize("mode", ["polynomial", "per_sample"])
def test_split_kernel_has_the_one_wave_kernels_bits(gpu_ctx, mode):
    """Launches of at most 4096 pairs sample a pair with 4 or 2 waves (pairs_split_kernel, LDS-ordered sums); larger launches
    with one (pairs_kernel).  The same pairs through both -- all 4950 pairs of 100 views in one launch against index lists and
    sub-ranges of 50 ... 4000 of them -- must give the same bits."""
    import epipolarconsistency_amd as E
    n = 100
    Ps, base, dtrs = _scan(gpu_ctx, n, B=64, seed=11)
    m = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setSampling(mode).setSmallEval(False).setRecordReuse(False)
    n_pairs = n * (n - 1) // 2
    total, vals = m.evaluate_range(0, n_pairs, want_pairs=True)  # 4950 pairs: one wave per pair
    assert np.isfinite(vals).all()
    for first, count in ((0, 50), (17, 1500), (1000, 1792), (2000, 1793), (900, 4000), (4900, 50)):
        s, v = m.evaluate_range(first, count, want_pairs=True)
        assert np.array_equal(v, vals[first:first + count]), (mode, first, count)
    rng = np.random.default_rng(1)
    for L in (1, 77, 1792, 1793, 4096):
        sel = np.sort(rng.choice(n_pairs, size=L, replace=False))
        idx = np.array([(*E.get_ij(int(q), n), *E.get_ij(int(q), n)) for q in sel], np.int32)
        out = np.empty(L, np.float32)
        m.evaluate(idx, out)
        assert np.array_equal(out, vals[sel]), (mode, L)
    m.close()
    for d in base:
        d.close()
