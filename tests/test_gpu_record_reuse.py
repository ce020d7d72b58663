"""Record reuse (ecc_metric_set_record_reuse, default on; not in the reference).

The reference's optimisation problems overwrite ONE view's matrix per cost-function call and evaluate all pairs again
(ref: Gui/SingleImageMotion.h:84-90).  A pair's geometry record (the reference's K01 array, ref: ...RadonIntermediate.cu:13-67)
depends on the pair's two matrices and the parameters only, so the library keeps the records of the last evaluation,
refits only the pairs with a changed matrix, computes E1 of the changed views on the host and does not launch e1_kernel.
Every pair is still sampled.  The contract tested here: results (mean, every pair value, cost images, E1 on the device)
are BIT-IDENTICAL to a metric with the switch off, whatever the sequence of calls."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _moved(Ps, views, k=1.0):
    import epipolarconsistency_amd as E
    out = list(Ps)
    for v in views:
        out[v] = out[v] @ E.geometry.rigid_transform(tx=0.7 * k, ty=-0.3 * k, rz=0.01 * k, ry=0.004 * (v + 1))
    return out


def _pair(gpu_ctx, Ps, dtrs):
    import epipolarconsistency_amd as E
    on = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setRecordReuse(True, always=True)  # (the default leaves small ranges alone)
    off = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setRecordReuse(False)
    return on, off


def _same(on, off, n_pairs):
    a, va = on.evaluate_range(0, n_pairs, want_pairs=True)
    b, vb = off.evaluate_range(0, n_pairs, want_pairs=True)
    assert a == b and np.array_equal(va, vb)
    assert on.evaluate() == off.evaluate()
    for x, y in zip(on.debug_geometry(), off.debug_geometry()):
        assert np.array_equal(x, y)
    return a


def test_sequences_bit_identical(gpu_ctx, small_scan):
    import epipolarconsistency_amd as E
    s = small_scan
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in s["dtrs"]]
    on, off = _pair(gpu_ctx, s["Ps"], dtrs)
    n_pairs = 28
    first = _same(on, off, n_pairs)
    seen = {first}
    Ps = list(s["Ps"])
    # one view, none, two views, view 0 (the automatic object radius follows it: everything is refitted), three views
    # (more than a quarter: everything), one view again
    for k, views in enumerate([[3], [], [1, 6], [0], [0, 2, 4], [7]]):
        Ps = _moved(Ps, views, k + 1.0)
        on.setProjectionMatrices(Ps)
        off.setProjectionMatrices(Ps)
        v = _same(on, off, n_pairs)
        if views:
            assert v not in seen
        seen.add(v)
    # moved and moved back: the first value again, bit for bit
    on.setProjectionMatrices(s["Ps"]); off.setProjectionMatrices(s["Ps"])
    assert _same(on, off, n_pairs) == first
    on.setProjectionMatrices(_moved(s["Ps"], [5])); off.setProjectionMatrices(_moved(s["Ps"], [5]))
    assert _same(on, off, n_pairs) != first
    on.setProjectionMatrices(s["Ps"]); off.setProjectionMatrices(s["Ps"])
    assert _same(on, off, n_pairs) == first
    # cost images: untouched entries survive, touched ones identical
    for views in ([], [2], [4]):
        P1 = _moved(s["Ps"], views, 0.3)
        ca, cb = np.full((8, 8), 2.0, np.float32), np.full((8, 8), 2.0, np.float32)
        assert on.setProjectionMatrices(P1).evaluate(ca) == off.setProjectionMatrices(P1).evaluate(cb)
        assert np.array_equal(ca, cb) and ca[3, 3] == 2.0 and ca[0, 1] == 2.0 and ca[1, 0] != 2.0  # index i + j * n, i < j
    # parameters, sampling modes: records of other parameters are never used
    for change in (lambda m: m.setObjectRadius(25.0), lambda m: m.setdKappa(0.004), lambda m: m.useCorrelation(True),
                   lambda m: m.useCorrelation(False).setObjectRadius(0.0).setdKappa(0.0), lambda m: m.setSampling("per_sample"),
                   lambda m: m.setSampling("reference"), lambda m: m.setSampling("auto"), lambda m: m.setSampling("polynomial")):
        change(on)
        change(off)
        _same(on, off, n_pairs)
        P1 = _moved(s["Ps"], [2], 0.5)
        on.setProjectionMatrices(P1); off.setProjectionMatrices(P1)
        _same(on, off, n_pairs)
        on.setProjectionMatrices(s["Ps"]); off.setProjectionMatrices(s["Ps"])
    assert _same(on, off, n_pairs) == first
    # index lists and image pairs in between use the record array: the kept records are dropped, not misused
    idx = [(0, 5, 0, 5), (2, 3, 2, 3)]
    assert on.evaluate(idx) == off.evaluate(idx)
    on.setProjectionMatrices(_moved(s["Ps"], [6], 0.2)); off.setProjectionMatrices(_moved(s["Ps"], [6], 0.2))
    _same(on, off, n_pairs)
    a, b = on.evaluateForImagePair(1, 6), off.evaluateForImagePair(1, 6)
    assert a[0] == b[0] and all(np.array_equal(a[1][k], b[1][k]) for k in a[1])
    _same(on, off, n_pairs)
    # the pose-delta mode on top of it
    on.setIncremental(True)
    for views in ([1], [], [4, 5]):
        P1 = _moved(s["Ps"], views, 0.9)
        assert on.setProjectionMatrices(P1).evaluate() == off.setProjectionMatrices(P1).evaluate()
    on.setIncremental(False)
    # switching off and on again
    on.setRecordReuse(False)
    _same(on, off, n_pairs)
    on.setRecordReuse(True, always=True)
    _same(on, off, n_pairs)
    on.setProjectionMatrices(_moved(s["Ps"], [3], 0.1)); off.setProjectionMatrices(_moved(s["Ps"], [3], 0.1))
    _same(on, off, n_pairs)
    on.close()
    off.close()


def test_shards_and_ranges(gpu_ctx, small_scan):
    """Ranges of the pair triangle (what one rank of a sharded evaluation calls): the changed view's pairs may all lie
    outside the range (nothing refitted, nothing launched for E1), and a different range starts from scratch."""
    import epipolarconsistency_amd as E
    s = small_scan
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in s["dtrs"]]
    on, off = _pair(gpu_ctx, s["Ps"], dtrs)
    ranges = [(0, 7), (7, 6), (13, 15), (0, 28), (5, 0)]  # (0, 7) = the pairs of view 0 only: every other view's move leaves L = 0 ... except through view 0's partners
    for first, count in ranges:
        for views in ([], [7], [1], [7], [2, 3], []):
            P1 = _moved(s["Ps"], views, 0.4 + 0.1 * len(views))
            on.setProjectionMatrices(P1); off.setProjectionMatrices(P1)
            a, va = on.evaluate_range(first, count, want_pairs=True)
            b, vb = off.evaluate_range(first, count, want_pairs=True)
            assert a == b and np.array_equal(va, vb), (first, count, views)
    # pairs (6, 7) only: moving view 1 touches no pair of the range; afterwards a full evaluation must see view 1's new E1
    P1 = _moved(s["Ps"], [1], 0.77)
    for m in (on, off):
        m.setProjectionMatrices(s["Ps"]).evaluate_range(27, 1)
        m.setProjectionMatrices(P1)
    assert on.evaluate_range(27, 1) == off.evaluate_range(27, 1)
    assert on.evaluate() == off.evaluate()
    for x, y in zip(on.debug_geometry(), off.debug_geometry()):
        assert np.array_equal(x, y)
    on.close()
    off.close()


def test_asynchronous_ranges_cycle_the_list_buffers(gpu_ctx, small_scan):
    import torch
    import epipolarconsistency_amd as E
    s = small_scan
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in s["dtrs"]]
    on, off = _pair(gpu_ctx, s["Ps"], dtrs)
    dev = torch.device("cuda", 0)
    sums = torch.zeros(12, dtype=torch.float64, device=dev)
    want = []
    for k in range(12):  # no synchronisation between the calls
        P1 = _moved(s["Ps"], [k % 8] if k % 3 else [], 0.1 * (k + 1))
        on.setProjectionMatrices(P1)
        on.evaluate_range_async(0, 28, sums[k:k + 1])
        want.append(off.setProjectionMatrices(P1).evaluate_range(0, 28))
    gpu_ctx.synchronize()
    torch.cuda.synchronize()
    assert sums.cpu().tolist() == want
    on.close()
    off.close()


def test_many_changed_views_take_the_wide_list_kernel(gpu_ctx):
    """200 views, 24 of them moved: 4 500 listed pairs, above the 8-lanes-per-fit limit of k01_kernel (4096)."""
    import epipolarconsistency_amd as E
    rng = np.random.default_rng(3)
    n, S, B = 200, 128, 64
    from epipolarconsistency_amd import synthetic
    Ps = synthetic.short_scan(n, S, S, 0.308 * 1024 / S)
    data = [rng.standard_normal((B, B)).astype(np.float32) for _ in range(4)]
    base = [E.RadonIntermediate.from_host(gpu_ctx, d, S, S) for d in data]
    dtrs = [base[v % 4] for v in range(n)]
    on, off = _pair(gpu_ctx, Ps, dtrs)
    n_pairs = n * (n - 1) // 2
    _same(on, off, n_pairs)
    moved = list(range(5, 200, 8))[:24]
    P1 = _moved(Ps, moved, 0.6)
    on.setProjectionMatrices(P1); off.setProjectionMatrices(P1)
    _same(on, off, n_pairs)
    P2 = _moved(P1, [100], 0.2)
    on.setProjectionMatrices(P2); off.setProjectionMatrices(P2)
    _same(on, off, n_pairs)
    on.close()
    off.close()
    for d in base:
        d.close()


def test_group_with_reuse_on_every_rank(gpu_ctx, small_scan, monkeypatch):
    """The single-process group with record reuse on every rank (ECC_RECORD_REUSE=2: for every size -- the library's
    default leaves ranges below 8192 pairs alone); compare with reuse off."""
    import epipolarconsistency_amd as E
    s = small_scan
    monkeypatch.setenv("ECC_RECORD_REUSE", "2")
    g = E.Group([0, 0, 0])
    gd = g.compute_batch(s["imgs"], s["n_alpha"], s["n_t"])
    gm = E.GroupMetricRadonIntermediate(g, s["Ps"], gd)
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in s["dtrs"]]
    off = E.MetricRadonIntermediate(gpu_ctx, s["Ps"], dtrs).setRecordReuse(False)
    bnd = off.balanced_shards(3)  # the group's shard bounds: fixed at creation from the first matrices
    for views in ([], [3], [3], [0], [1, 2], []):
        P1 = _moved(s["Ps"], views, 0.35)
        a = gm.setProjectionMatrices(P1).evaluate()
        off.setProjectionMatrices(P1)
        parts = [off.evaluate_range(bnd[r], bnd[r + 1] - bnd[r]) for r in range(3)]
        assert a == (parts[0] + parts[1] + parts[2]) / 28, views
    gm.close()
    off.close()
    g.close()


def test_more_views_than_the_skip_set_holds(gpu_ctx):
    """520 views: the two-stream form's skip set is a 512-bit kernel argument, so the refit runs in front of the
    all-pairs launch on the context's stream instead -- same bits."""
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import synthetic
    rng = np.random.default_rng(8)
    n, S, B = 520, 96, 48
    Ps = synthetic.short_scan(n, S, S, 0.308 * 1024 / S)
    base = [E.RadonIntermediate.from_host(gpu_ctx, rng.standard_normal((B, B)).astype(np.float32), S, S) for _ in range(3)]
    dtrs = [base[v % 3] for v in range(n)]
    on, off = _pair(gpu_ctx, Ps, dtrs)
    n_pairs = n * (n - 1) // 2
    assert on.evaluate() == off.evaluate()
    for views in ([515], [3, 519], []):
        P1 = _moved(Ps, views, 0.4)
        a, va = on.setProjectionMatrices(P1).evaluate_range(0, n_pairs, want_pairs=True)
        b, vb = off.setProjectionMatrices(P1).evaluate_range(0, n_pairs, want_pairs=True)
        assert a == b and np.array_equal(va, vb), views
    on.close()
    off.close()
    for d in base:
        d.close()


def test_published_scalar_equals_the_copied_one(gpu_ctx, small_scan):
    """sharding.distributed_evaluate(publish=True): the (all-reduced) sum comes back through the metric's pinned result
    slot instead of tensor.item(); same value, also over several calls in a row."""
    import torch
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import sharding
    s = small_scan
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in s["dtrs"]]
    m = E.MetricRadonIntermediate(gpu_ctx, s["Ps"], dtrs).setRecordReuse(True, always=True)
    sum_t = torch.zeros(1, dtype=torch.float64, device=torch.device("cuda", 0))
    for k, views in enumerate(([], [2], [2], [5, 6], [])):
        P1 = _moved(s["Ps"], views, 0.2 * (k + 1))
        m.setProjectionMatrices(P1)
        a = sharding.distributed_evaluate(m, 8, sum_t, 0, 1, publish=True)
        gpu_ctx.synchronize()
        assert a == float(sum_t.item()) / 28
        assert a == sharding.distributed_evaluate(m, 8, sum_t, 0, 1, publish=False) == m.evaluate()
    m.close()


def test_geometry_users_between_two_sets_drop_the_kept_records(gpu_ctx, small_scan):
    """Round-3 advisor finding: evaluateForImagePair / debug_geometry run E1 for ALL views of the currently staged
    matrices.  After set(P + e_v); <one of them>; set(P + e_w) the device holds E1((P + e_v)_v) while the kept records
    belong to P -- the refit of pair (v, w) must not read that stale geometry.  Also the variant with no view changed
    relative to the kept matrices: set(P + e_v); debug_geometry; set(P); evaluate; then users of the device geometry."""
    import epipolarconsistency_amd as E
    s = small_scan
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in s["dtrs"]]
    n_pairs = 28
    for user in ("debug_geometry", "image_pair"):
        on, off = _pair(gpu_ctx, s["Ps"], dtrs)
        _same(on, off, n_pairs)
        for v, w in ((2, 5), (6, 1), (3, 3)):
            Pv = _moved(s["Ps"], [v], 0.8)
            Pw = _moved(s["Ps"], [w], 0.6)
            for m in (on, off):
                m.setProjectionMatrices(Pv)
                if user == "debug_geometry":
                    m.debug_geometry()
                else:
                    m.evaluateForImagePair(min(v, 7 - v), 7)
                m.setProjectionMatrices(Pw)
            a, va = on.evaluate_range(0, n_pairs, want_pairs=True)
            b, vb = off.evaluate_range(0, n_pairs, want_pairs=True)
            assert a == b and np.array_equal(va, vb), (user, v, w)
        # no view changed relative to the kept matrices, but the device geometry was overwritten in between
        Pv = _moved(s["Ps"], [4], 0.5)
        for m in (on, off):
            m.setProjectionMatrices(s["Ps"]).evaluate()
            m.setProjectionMatrices(Pv)
            m.debug_geometry()
            m.setProjectionMatrices(s["Ps"])
        assert on.evaluate() == off.evaluate()
        idx = np.array([[4, 6, 4, 6], [1, 4, 1, 4], [0, 7, 0, 7]], np.int32)
        assert on.evaluate(idx) == off.evaluate(idx)
        for x, y in zip(on.debug_geometry(), off.debug_geometry()):
            assert np.array_equal(x, y)
        (ea, da), (eb, db) = on.evaluateForImagePair(4, 6), off.evaluateForImagePair(4, 6)
        assert ea == eb
        for k in da:
            assert np.array_equal(da[k], db[k]), k
        on.close()
        off.close()


def test_pipelined_publish_and_wait_keep_the_fork_event(gpu_ctx):
    """Round-5 advisor finding: ecc_metric_wait_scalar(k) proves only that the work queued BEFORE publish(k) has run.  In the
    pipelined order async(k), publish(k), async(k + 1), wait(k), async(k + 2) the evaluation k + 1 may still be running on
    both streams when k + 2's side-stream refit starts: the metric must not be declared quiet (which would drop the fork
    event in front of the refit).  Sizes that take the two-stream form (the refit's k01 rewrites records, its list launch
    rewrites value slots the pending launches read), every published value against a metric without reuse."""
    import torch
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import synthetic
    rng = np.random.default_rng(17)
    n, S, B = 150, 128, 48   # 11 175 pairs
    Ps = synthetic.short_scan(n, S, S, 0.308 * 1024 / S)
    base = [E.RadonIntermediate.from_host(gpu_ctx, rng.standard_normal((B, B)).astype(np.float32), S, S) for _ in range(5)]
    dtrs = [base[v % 5] for v in range(n)]
    on = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setRecordReuse(True, always=True)
    off = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setRecordReuse(False)
    n_pairs = n * (n - 1) // 2
    dev = torch.device("cuda", 0)
    sums = torch.zeros(40, dtype=torch.float64, device=dev)
    pairs = [torch.zeros(n_pairs, dtype=torch.float32, device=dev) for _ in range(2)]
    poses = [_moved(Ps, [(7 * k + 3) % n, (k * k) % n] if k % 3 else [(5 * k) % n], 0.05 * (k + 1)) for k in range(40)]
    want = [off.setProjectionMatrices(P).evaluate_range(0, n_pairs) for P in poses]
    on.setProjectionMatrices(poses[0]).evaluate_range(0, n_pairs)  # kept records: the calls below take the refit
    got = []
    on.setProjectionMatrices(poses[0])
    on.evaluate_range_async(0, n_pairs, sums[0:1], pairs[0])
    on.publish_scalar(sums[0:1])
    for k in range(1, 40):
        on.setProjectionMatrices(poses[k])
        on.evaluate_range_async(0, n_pairs, sums[k:k + 1], pairs[k & 1])   # queued behind publish(k - 1) ...
        got.append(on.wait_scalar())                                       # ... and possibly still running here
        on.publish_scalar(sums[k:k + 1])
    got.append(on.wait_scalar())
    gpu_ctx.synchronize()
    torch.cuda.synchronize()
    assert got == want
    assert sums.cpu().tolist() == want
    # after a wait with nothing queued behind the publish the synchronous path continues unharmed
    assert on.setProjectionMatrices(poses[3]).evaluate_range(0, n_pairs) == want[3]
    on.close(); off.close()
    for d in base:
        d.close()
