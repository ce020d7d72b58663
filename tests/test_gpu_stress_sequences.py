"""Randomised interleavings of everything that keeps state between evaluations -- kept pair records and the two-stream refit
(ecc_metric_set_record_reuse), kept pair values (ecc_metric_set_incremental), the tracked device geometry, the one-launch
path for small evaluations (ecc_metric_set_small_eval) -- against a metric with all of it switched off.  Every result (means,
pair values, cost images, E1 on the device, image-pair curves) must be BIT-IDENTICAL, whatever the sequence of calls:
the reference's callers overwrite one view's matrix per cost-function call and evaluate again (ref:
Gui/SingleImageMotion.h:84-90, tools/FluoroTracking/FluoroTracking.cpp:179-211), and nothing that is skipped may show.
n = 20 / 64 / 130 / 520 views cross the 512-pair (sampling mode), 4096-pair (one launch, record reuse), 8192-pair
(two-stream refit) and 512-view (skip mask) thresholds."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sequence(rng, n, m, ref, P0, slabs, torch, E, ctx, n_ops, log):
    n_pairs = n * (n - 1) // 2
    dev = slabs.device
    P = P0.copy()
    sum_t = torch.zeros(1, dtype=torch.float64, device=dev)
    modes = ["auto", "polynomial", "per_sample"] + (["reference"] if n <= 20 else [])

    def both(f):
        f(m)
        f(ref)

    def set_matrices(k):
        nonlocal P
        P = P.copy()
        for v in rng.choice(n, size=k, replace=False):
            T = E.geometry.rigid_transform(tx=float(rng.uniform(-1, 1)), ty=float(rng.uniform(-1, 1)), rz=float(rng.uniform(-.01, .01)),
                                           ry=float(rng.uniform(-.01, .01)))
            P[v] = (P[v].reshape(4, 3).T @ T).T.reshape(12)
        if rng.random() < 0.15:
            P = P0.copy()  # back to the start: "changed" relative to whatever is kept
        both(lambda q: q.setProjectionMatrices(P))

    for op_i in range(n_ops):
        op = rng.choice(["set1", "set1", "set1", "set0", "set2", "setmany", "all", "all", "all", "cost", "range", "range", "async",
                         "list", "list", "pair", "geo", "refresh", "params", "mode", "incr", "reuse", "small", "poses"])
        log.append(op)
        if op == "set0":
            set_matrices(0)
        elif op == "set1":
            set_matrices(1)
        elif op == "set2":
            set_matrices(2)
        elif op == "setmany":
            set_matrices(int(rng.integers(max(3, n // 6), n + 1)))
        elif op == "all":
            assert m.evaluate() == ref.evaluate(), log[-12:]
        elif op == "cost":
            ca = np.full((n, n), 2.5, np.float32)
            cb = ca.copy()
            assert m.evaluate(ca) == ref.evaluate(cb), log[-12:]
            assert np.array_equal(ca, cb), log[-12:]
        elif op == "range":
            first = int(rng.integers(0, n_pairs))
            count = int(rng.integers(0, n_pairs - first + 1)) if rng.random() < 0.5 else int(min(n_pairs - first, rng.integers(1, 600)))
            want = bool(rng.random() < 0.5)
            a, b = m.evaluate_range(first, count, want_pairs=want), ref.evaluate_range(first, count, want_pairs=want)
            if want:
                assert a[0] == b[0] and np.array_equal(a[1], b[1]), log[-12:]
            else:
                assert a == b, log[-12:]
        elif op == "async":
            first = int(rng.integers(0, n_pairs))
            count = int(rng.integers(1, n_pairs - first + 1))
            m.evaluate_range_async(first, count, sum_t)
            if rng.random() < 0.5:  # a second call queued behind it without any wait
                set_matrices(1)
                want2 = ref.evaluate_range(first, count)
                m.evaluate_range_async(first, count, sum_t)
                ctx.synchronize()
                assert float(sum_t.item()) == want2, log[-12:]
            else:
                ctx.synchronize()
                assert float(sum_t.item()) == ref.evaluate_range(first, count), log[-12:]
        elif op == "list":
            L = int(rng.choice([1, 3, 40, 200, 513, 700]))
            ab = rng.integers(0, n, size=(L, 2))
            idx = np.stack([ab[:, 0], ab[:, 1], rng.integers(0, n, L) if rng.random() < 0.2 else ab[:, 0],
                            rng.integers(0, n, L) if rng.random() < 0.2 else ab[:, 1]], axis=1).astype(np.int32)
            oa, ob = np.empty(L, np.float32), np.empty(L, np.float32)
            a, b = m.evaluate(idx, oa), ref.evaluate(idx, ob)
            assert (a == b or (np.isnan(a) and np.isnan(b))) and np.array_equal(oa, ob, equal_nan=True), log[-12:]
        elif op == "pair":
            i, j = (int(v) for v in rng.choice(n, 2, replace=False))
            (ea, da), (eb, db) = m.evaluateForImagePair(i, j), ref.evaluateForImagePair(i, j)
            assert ea == eb and all(np.array_equal(da[k], db[k]) for k in da), log[-12:]
        elif op == "geo":
            for xa, xb in zip(m.debug_geometry(), ref.debug_geometry()):
                assert np.array_equal(xa, xb), log[-12:]
        elif op == "refresh":
            v = int(rng.integers(0, n))
            slabs[v % slabs.shape[0]].mul_(float(rng.uniform(0.9, 1.1)))  # on the context's stream (torch's current stream)
            both(lambda q: q.refreshRadonIntermediates())
        elif op == "params":
            r, dk = float(rng.choice([0.0, 0.0, 20.0, 60.0])), float(rng.choice([0.0, 0.0, 0.003]))
            both(lambda q: q.setObjectRadius(r).setdKappa(dk))
            if rng.random() < 0.15:
                c = bool(rng.random() < 0.5)
                both(lambda q: q.useCorrelation(c))
        elif op == "mode":
            md = str(rng.choice(modes))
            both(lambda q: q.setSampling(md))
        elif op == "incr":
            m.setIncremental(bool(rng.random() < 0.5))
        elif op == "reuse":
            k = int(rng.integers(0, 3))
            m.setRecordReuse(k > 0, always=k == 2)
        elif op == "small":
            m.setSmallEval(bool(rng.random() < 0.7))
        elif op == "poses":  # ecc_metric_evaluate_poses (two deep): the last pose stays the metric's current one
            poses = []
            for _ in range(int(rng.integers(1, 4))):
                Pq = P.copy()
                v = int(rng.integers(0, n))
                Pq[v] = (Pq[v].reshape(4, 3).T @ E.geometry.rigid_transform(tx=float(rng.uniform(-1, 1)), rz=float(rng.uniform(-.01, .01)))).T.reshape(12)
                poses.append(Pq)
            got = m.evaluate_poses(poses)
            want = np.array([ref.setProjectionMatrices(Pq).evaluate() for Pq in poses])
            assert np.array_equal(np.asarray(got), want, equal_nan=True), log[-12:]
            P = poses[-1]
    both(lambda q: q.useCorrelation(False).setObjectRadius(0.0).setdKappa(0.0))
    assert m.evaluate() == ref.evaluate(), log[-12:]


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("n,sequences,n_ops,seed", [(20, 220, 14, 1), (64, 150, 14, 2), (130, 100, 12, 3), (520, 40, 10, 4)])
def test_randomised_sequences_are_bit_identical_to_a_stateless_metric(gpu_ctx, n, sequences, n_ops, seed):
    import torch
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import synthetic
    rng = np.random.default_rng(seed)
    S, B = 128, 48
    dev = torch.device("cuda", gpu_ctx.device)
    Ps = synthetic.short_scan(n, S, S, 0.308 * 1024 / S)
    P0 = E.pack_projection_matrices(Ps)
    base = 7  # distinct Radon intermediates; view v samples number v % 7
    slabs = torch.zeros((base, E.slab_floats(B, B)), dtype=torch.float32, device=dev)  # owned here: "refresh" rescales them in place
    imgs = torch.from_numpy(np.stack([rng.uniform(0, 50, (S, S)).astype(np.float32) for _ in range(base)])).to(dev)
    keep = E.RadonIntermediate.compute_into(gpu_ctx, imgs, slabs, B, B)
    gpu_ctx.synchronize()
    dtrs = [keep[v % base] for v in range(n)]
    for s in range(sequences):
        m = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setSampling("auto")
        ref = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setSampling("auto").setRecordReuse(False).setSmallEval(False)
        log = ["sequence %d" % s]
        _sequence(rng, n, m, ref, P0, slabs, torch, E, gpu_ctx, n_ops, log)
        m.close()
        ref.close()
