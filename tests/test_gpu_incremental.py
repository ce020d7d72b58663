"""Pose-delta evaluation (ecc_metric_set_incremental / ecc_group_metric_set_incremental; not in the reference).

The reference's optimisation problems move ONE view per cost-function call and evaluate all pairs again
(ref: Gui/SingleImageMotion.h:84-90, Gui/Visualization.h:78-98).  With the opt-in mode the metric keeps the pair values
of its last evaluation on the device and re-evaluates only the pairs that contain a view whose matrix changed.  The
contract tested here: every result is BIT-IDENTICAL to the one a metric without the mode returns for the same call, the
number of recomputed pairs is what the matrices' differences say, and everything that changes a pair's value without
changing a matrix (parameters, sampling mode, range, refreshed Radon intermediates) drops the kept values."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _moved(Ps, views, k=1.0):
    import epipolarconsistency_amd as E
    out = list(Ps)
    for v in views:
        out[v] = out[v] @ E.geometry.rigid_transform(tx=0.7 * k, ty=-0.3 * k, rz=0.01 * k, ry=0.004 * (v + 1))
    return out


def test_single_metric_sequence_bit_identical(gpu_ctx, small_scan):
    import epipolarconsistency_amd as E
    s = small_scan
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in s["dtrs"]]
    full = E.MetricRadonIntermediate(gpu_ctx, s["Ps"], dtrs)
    inc = E.MetricRadonIntermediate(gpu_ctx, s["Ps"], dtrs).setIncremental(True)
    n_pairs = 28
    first_value = full.evaluate()
    assert inc.evaluate() == first_value and inc.last_evaluated_pairs() == n_pairs
    steps = [([3], 7), ([], 0), ([1, 6], 13), ([0, 2, 4], n_pairs)]  # (views moved in this step, pairs recomputed)
    Ps = list(s["Ps"])
    seen = {first_value}
    for k, (views, want_pairs) in enumerate(steps):
        Ps = _moved(Ps, views, k + 1.0)
        a = full.setProjectionMatrices(Ps).evaluate()
        b = inc.setProjectionMatrices(Ps).evaluate()
        assert a == b, (k, a, b)
        assert inc.last_evaluated_pairs() == want_pairs, (k, inc.last_evaluated_pairs())
        assert full.last_evaluated_pairs() == n_pairs
        if views:
            assert a not in seen
        seen.add(a)
    # back to the first pose: six views differ from the kept ones -> everything again, the first value again
    assert inc.setProjectionMatrices(s["Ps"]).evaluate() == first_value and inc.last_evaluated_pairs() == n_pairs
    # one view moved, then moved back: two incremental steps, the first value again bit for bit
    b1 = inc.setProjectionMatrices(_moved(s["Ps"], [5])).evaluate()
    assert inc.last_evaluated_pairs() == 7 and b1 != first_value
    assert inc.setProjectionMatrices(s["Ps"]).evaluate() == first_value and inc.last_evaluated_pairs() == 7
    # a cost image writes every pair: full evaluation, kept values untouched
    cost_a, cost_b = np.full((8, 8), 2.0, np.float32), np.full((8, 8), 2.0, np.float32)
    full.setProjectionMatrices(s["Ps"])
    assert full.evaluate(cost_a) == inc.evaluate(cost_b) and np.array_equal(cost_a, cost_b)
    assert inc.last_evaluated_pairs() == n_pairs
    assert inc.evaluate() == first_value and inc.last_evaluated_pairs() == 0
    # parameters and the sampling mode drop the kept values
    for change in (lambda m: m.setObjectRadius(25.0), lambda m: m.setdKappa(0.004), lambda m: m.useCorrelation(True),
                   lambda m: m.useCorrelation(False).setObjectRadius(0.0).setdKappa(0.0), lambda m: m.setSampling("per_sample"),
                   lambda m: m.setSampling("reference"), lambda m: m.setSampling("polynomial")):
        change(full)
        change(inc)
        assert full.evaluate() == inc.evaluate() and inc.last_evaluated_pairs() == n_pairs
        Ps1 = _moved(s["Ps"], [2], 0.5)
        assert full.setProjectionMatrices(Ps1).evaluate() == inc.setProjectionMatrices(Ps1).evaluate()
        assert inc.last_evaluated_pairs() == 7
        full.setProjectionMatrices(s["Ps"])
        inc.setProjectionMatrices(s["Ps"])
    assert inc.evaluate() == first_value
    # index lists and single pairs do not touch the kept values
    idx = [(0, 5, 0, 5), (2, 3, 2, 3)]
    assert inc.evaluate(idx) == full.evaluate(idx)
    assert inc.evaluate() == first_value and inc.last_evaluated_pairs() == 0
    # switching the mode off and on again starts from scratch
    inc.setIncremental(False)
    assert inc.evaluate() == first_value and inc.last_evaluated_pairs() == n_pairs
    inc.setIncremental(True)
    assert inc.evaluate() == first_value and inc.last_evaluated_pairs() == n_pairs
    full.close()
    inc.close()


def test_ranges_keep_their_own_values(gpu_ctx, small_scan):
    import epipolarconsistency_amd as E
    s = small_scan
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in s["dtrs"]]
    full = E.MetricRadonIntermediate(gpu_ctx, s["Ps"], dtrs)
    inc = E.MetricRadonIntermediate(gpu_ctx, s["Ps"], dtrs).setIncremental(True)
    first, count = 5, 20
    a, va = full.evaluate_range(first, count, want_pairs=True)
    b, vb = inc.evaluate_range(first, count, want_pairs=True)
    assert a == b and np.array_equal(va, vb) and inc.last_evaluated_pairs() == count
    Ps = _moved(s["Ps"], [4])
    a, va = full.setProjectionMatrices(Ps).evaluate_range(first, count, want_pairs=True)
    b, vb = inc.setProjectionMatrices(Ps).evaluate_range(first, count, want_pairs=True)
    assert a == b and np.array_equal(va, vb)
    # pairs of view 4 inside [5, 25) of the get_ij order
    in_range = sum(1 for q in range(first, first + count) if 4 in E.get_ij(q, 8))
    assert 0 < in_range < 7 and inc.last_evaluated_pairs() == in_range
    # another range: evaluated in full, and it replaces the kept one
    assert inc.evaluate_range(0, 28) == full.evaluate_range(0, 28) and inc.last_evaluated_pairs() == 28
    assert inc.evaluate_range(first, count) == a and inc.last_evaluated_pairs() == count
    # an empty range
    assert inc.evaluate_range(7, 0) == 0.0
    full.close()
    inc.close()


def test_mode_of_the_full_range_and_refreshed_dtrs(gpu_ctx):
    """34 views = 561 pairs: in ECC_SAMPLING_AUTO the full range runs on the polynomial path, and so must the 33 pairs of
    a moved view although an evaluation of 33 pairs by itself would use the reference arithmetic.  Then a recomputed dtr."""
    import torch
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import synthetic
    n, S, B = 34, 64, 48
    Ps = synthetic.short_scan(n, S, S, 4.9)
    dev = torch.device("cuda", gpu_ctx.device)
    imgs = synthetic.projections_torch(Ps, S, S, synthetic.sphere_phantom(extent_mm=30, rmin=8, rmax=25), dev)
    slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
    dtrs = E.RadonIntermediate.compute_into(gpu_ctx, imgs, slabs, B, B)
    gpu_ctx.synchronize()
    full = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setSampling("auto")
    inc = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setSampling("auto").setIncremental(True)
    assert full.evaluate() == inc.evaluate() and inc.last_evaluated_pairs() == 561
    Ps1 = _moved(Ps, [17])
    a, b = full.setProjectionMatrices(Ps1).evaluate(), inc.setProjectionMatrices(Ps1).evaluate()
    assert a == b and inc.last_evaluated_pairs() == 33
    ref33 = E.MetricRadonIntermediate(gpu_ctx, Ps1, dtrs).setSampling("reference")
    poly33 = E.MetricRadonIntermediate(gpu_ctx, Ps1, dtrs).setSampling("polynomial")
    idx = [(min(17, u), max(17, u)) * 2 for u in range(n) if u != 17]
    assert ref33.evaluate(idx) != poly33.evaluate(idx)  # the two modes differ on these pairs, so the check above is sharp
    # the Radon intermediate of view 9 recomputed in place from another image: refresh, then everything is re-evaluated
    imgs[9] = imgs[9] * 1.5 + 0.25
    E.RadonIntermediate.compute_into(gpu_ctx, imgs[9:10], slabs[9:10], B, B)
    full.refreshRadonIntermediates(9, 1)
    inc.refreshRadonIntermediates(9, 1)
    a2, b2 = full.evaluate(), inc.evaluate()
    assert a2 == b2 != a and inc.last_evaluated_pairs() == 561
    for m in (full, inc, ref33, poly33):
        m.close()


@pytest.mark.parametrize("ranks", [1, 3])
def test_group_shards_keep_their_values(gpu_ctx, small_scan, ranks):
    import epipolarconsistency_amd as E
    s = small_scan
    g = E.Group([0] * ranks)
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in s["dtrs"]]
    full = E.GroupMetricRadonIntermediate(g, s["Ps"], dtrs)
    inc = E.GroupMetricRadonIntermediate(g, s["Ps"], dtrs).setIncremental(True)
    first_value = full.evaluate()
    assert inc.evaluate() == first_value
    Ps = list(s["Ps"])
    for k, views in enumerate(([3], [], [0], [1, 6], [0, 2, 4, 5, 7])):
        Ps = _moved(Ps, views, k + 1.0)
        assert full.setProjectionMatrices(Ps).evaluate() == inc.setProjectionMatrices(Ps).evaluate(), (k, views)
    assert inc.setProjectionMatrices(s["Ps"]).evaluate() == first_value
    # the recomputed pairs of all ranks together are the moved view's pairs
    inc.setProjectionMatrices(_moved(s["Ps"], [6])).evaluate()
    assert inc.last_evaluated_pairs() == 7 and full.last_evaluated_pairs() == 28
    inc.close()
    full.close()
    g.close()
