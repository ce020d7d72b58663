"""The four sampling modes of the pair kernel (include/ecc_hip.h, ecc_metric_set_sampling) against the oracle on the
small synthetic scan: what each mode promises for single pair values and for the mean.

  reference   the CPU path's arithmetic, operation for operation: every pair value to float rounding of its sum;
  auto        = reference up to 512 pairs per evaluation, polynomial above;
  polynomial / per_sample   the throughput paths: mean 1e-5, single pairs at the fp32 noise floor (2e-4 here).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _metric(gpu_ctx, s, filt=0):
    import epipolarconsistency_amd as E
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"], filter=filt) for d in s["dtrs"]]
    return E.MetricRadonIntermediate(gpu_ctx, s["Ps"], dtrs), dtrs


def test_reference_mode_reproduces_every_pair(gpu_ctx, oracle_mod, small_scan):
    s = small_scan
    m, _ = _metric(gpu_ctx, s)
    want = oracle_mod.evaluate_all(s["Ps"], s["dtrs"], s["n_u"], s["n_v"], want_K01=True)
    for mode in ("reference", "auto"):
        m.setSampling(mode)
        cost = np.full((8, 8), -2.0, np.float32)
        mean = m.evaluate(cost)
        total, vals = m.evaluate_range(0, 28, want_pairs=True)
        np.testing.assert_allclose(vals, want["pairs"], rtol=1e-6)
        assert np.mean(vals == want["pairs"]) > 0.8  # bit-identical for most pairs (float64 sum order aside)
        assert abs(mean - want["mean"]) <= 1e-7 * want["mean"] and total / 28 == mean
        iu = np.triu_indices(8, 1)
        assert np.array_equal(cost[iu[1], iu[0]], vals) and np.all(cost[iu] == -2.0)
    # K01 of the pair-geometry kernel: the oracle's expressions, bit for bit except the two angles that go through
    # the device's float64 asin / atan2
    K = m.debug_K01(0, 28)
    np.testing.assert_allclose(K, want["K01s"], rtol=3e-7, atol=1e-12)
    m.close()


def test_modes_agree_on_the_mean(gpu_ctx, oracle_mod, small_scan):
    s = small_scan
    m, _ = _metric(gpu_ctx, s)
    want = oracle_mod.evaluate_all(s["Ps"], s["dtrs"], s["n_u"], s["n_v"])
    for mode, pair_tol in (("polynomial", 2e-4), ("per_sample", 2e-4), ("reference", 1e-6)):
        total, vals = m.setSampling(mode).evaluate_range(0, 28, want_pairs=True)
        assert abs(total / 28 - want["mean"]) <= 1e-5 * want["mean"], mode
        np.testing.assert_allclose(vals, want["pairs"], rtol=pair_tol, err_msg=mode)
    with pytest.raises(Exception):
        m.setSampling(7)
    m.close()


def test_reference_mode_variants(gpu_ctx, oracle_mod, small_scan):
    """Plain (non-derivative) dtrs, user dkappa / radius, useCorrelation and swapped index tuples in reference mode."""
    s = small_scan
    m, _ = _metric(gpu_ctx, s, filt=2)  # Filter::None -> no sign flip on the folded branch
    m.setSampling("reference")
    want = oracle_mod.evaluate_all(s["Ps"], s["dtrs"], s["n_u"], s["n_v"], is_derivative=False)
    total, vals = m.evaluate_range(0, 28, want_pairs=True)
    np.testing.assert_allclose(vals, want["pairs"], rtol=1e-6)
    m.close()
    m, _ = _metric(gpu_ctx, s)
    m.setSampling("reference").setObjectRadius(25.0).setEpipolarPlaneStep(0.004)
    want = oracle_mod.evaluate_all(s["Ps"], s["dtrs"], s["n_u"], s["n_v"], object_radius_mm=25.0, dkappa=0.004)
    total, vals = m.evaluate_range(0, 28, want_pairs=True)
    np.testing.assert_allclose(vals, want["pairs"], rtol=1e-6)
    m.setObjectRadius(0.0).setEpipolarPlaneStep(0.0)
    idx = np.array([[5, 2, 5, 2], [1, 6, 4, 2], [3, 3, 3, 3]], np.int32)  # reversed, crossed, degenerate (same view)
    want = oracle_mod.evaluate_pairs(s["Ps"], s["dtrs"], s["n_u"], s["n_v"], idx)
    out = np.zeros(3, np.float32)
    m.evaluate(idx, out)
    np.testing.assert_allclose(out, want["pairs"], rtol=1e-6, atol=0)
    assert out[2] == 0.0
    oracle_mod.set_use_corr(True)
    try:
        want = oracle_mod.evaluate_all(s["Ps"], s["dtrs"], s["n_u"], s["n_v"])
    finally:
        oracle_mod.set_use_corr(False)
    total, vals = m.useCorrelation(True).evaluate_range(0, 28, want_pairs=True)
    np.testing.assert_allclose(vals, want["pairs"], rtol=2e-5, atol=1e-7)  # 1 - cc: cancellation of a float near 1
    m.close()


def test_row_quad_copies_give_the_same_bits(gpu_ctx, small_scan):
    """Row-quad copies (ecc_ctx_set_quad_copies on the context before the metric is created; built by default while they fit):
    the pairs with kappa_max > pi/4 on the per-sample path, and the exact part of the pairs whose range exceeds the
    polynomials', sample them instead of the row-paired copies -- the same taps, the same arithmetic, identical values;
    also after refreshRadonIntermediates()."""
    import epipolarconsistency_amd as E
    s = small_scan
    gpu_ctx.setQuadCopies("off")
    try:
        m0, _ = _metric(gpu_ctx, s)
        gpu_ctx.setQuadCopies("on")
        m1, dtrs1 = _metric(gpu_ctx, s)
    finally:
        gpu_ctx.setQuadCopies("auto")
    K = m0.debug_K01(0, 28)
    assert (K[:, 15] > np.pi / 4).sum() >= 3, "the scan needs pairs with kappa_max > pi/4 for this test"
    for mode in ("per_sample", "polynomial"):
        a = m0.setSampling(mode).evaluate_range(0, 28, want_pairs=True)
        b = m1.setSampling(mode).evaluate_range(0, 28, want_pairs=True)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]), mode
    from epipolarconsistency_amd import _lib
    with pytest.raises(E.EccError):  # only ECC_QUAD_COPIES_AUTO / _OFF / _ON
        _lib.check(_lib.lib().ecc_ctx_set_quad_copies(gpu_ctx._h, 7))
    m1.refreshRadonIntermediates()
    b = m1.setSampling("per_sample").evaluate_range(0, 28, want_pairs=True)
    a = m0.setSampling("per_sample").evaluate_range(0, 28, want_pairs=True)
    assert np.array_equal(a[1], b[1])
    m0.close()
    m1.close()


def test_pairs_through_the_object_mix_polynomial_and_exact_samples(gpu_ctx, oracle_mod):
    """Pairs whose kappa range goes beyond the polynomials' 0.98 rad (ecc_layout.h: ecc_kappa_fit; kappa_max = pi/2 when the
    baseline passes through the object) take the inner samples from the polynomials and the rest from the exact loop.  On a
    short scan with many such pairs: they do carry polynomials; their values agree with the per-sample path and with the
    oracle like everybody else's; every kernel form gives the bits of the one-wave kernel (eight / four / two waves per pair,
    the small-evaluation kernel, row-quad copies for the exact part); the correlation variant agrees with the oracle."""
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import synthetic
    n, S, B = 72, 128, 96
    rng = np.random.default_rng(21)
    Ps = synthetic.short_scan(n, S, S, 0.308 * 1024 / S)
    host = [rng.standard_normal((B, B)).astype(np.float32) for _ in range(4)]
    base = [E.RadonIntermediate.from_host(gpu_ctx, h, S, S) for h in host]
    dtrs = [base[v % 4] for v in range(n)]
    n_pairs = n * (n - 1) // 2
    gpu_ctx.setQuadCopies("off")
    try:
        m = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setSmallEval(False).setRecordReuse(False)
    finally:
        gpu_ctx.setQuadCopies("auto")
    K = m.debug_K01(0, n_pairs)
    recs = m.setSampling("polynomial").debug_polynomials(0, n_pairs)
    far = np.flatnonzero(K[:, 15] > 0.98 + 1e-3)
    mixed = np.array([q for q in far if recs[q]["poly_ok"]], np.int64)
    assert len(far) >= 60 and len(mixed) >= 0.8 * len(far), (len(far), len(mixed))
    for q in mixed[:10]:
        assert abs(recs[q]["x_scale"] * np.float32(0.98) - 1) < 1e-6
    # one wave per pair, all pairs (2556 <= 4096 would take the split kernel: ask for ranges above that with a repeated list below)
    total_p, vp = m.evaluate_range(0, n_pairs, want_pairs=True)
    total_s, vs = m.setSampling("per_sample").evaluate_range(0, n_pairs, want_pairs=True)
    assert np.isfinite(vp).all() and (vp[mixed] > 0).all()
    rel = np.abs(vp - vs) / np.maximum(np.abs(vs), 1e-30)
    assert rel[mixed].max() < 2e-3 and abs(total_p - total_s) < 1e-5 * abs(total_s) * max(1.0, 30 / np.sqrt(n_pairs)), (rel[mixed].max(), total_p, total_s)
    assert not np.array_equal(vp[mixed], vs[mixed])  # (they are on a different path now)
    idx_all = np.array([(*E.get_ij(int(q), n), *E.get_ij(int(q), n)) for q in range(n_pairs)], np.int32)
    dt_host = [host[v % 4] for v in range(n)]
    sub = mixed[:: max(1, len(mixed) // 24)][:24]
    want = oracle_mod.evaluate_pairs(Ps, dt_host, S, S, idx_all[sub])
    got = vp[sub]
    assert np.abs(got - want["pairs"]).max() < 2e-3 * np.abs(want["pairs"]).max(), (got, want["pairs"])
    # every launch form against the 4-pairs-per-workgroup, one-wave-per-pair kernel: a list of > 4096 entries goes through it
    m.setSampling("polynomial")
    big = np.tile(mixed, 4096 // len(mixed) + 2)
    out_big = np.empty(len(big), np.float32)
    m.evaluate(idx_all[big], out_big)
    for L in (1, 5, 700, 2000, 4000):  # eight, eight, eight, four, two waves per pair
        sel = big[:L]
        out = np.empty(L, np.float32)
        m.evaluate(idx_all[sel], out)
        assert np.array_equal(out, out_big[:L]), L
    assert np.array_equal(out_big[:len(mixed)], vp[mixed])  # (the all-pairs launch of 2556 pairs used two waves per pair)
    m_small = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setSampling("polynomial")  # small evaluations: one launch (small_eval_kernel)
    out = np.empty(300, np.float32)
    m_small.evaluate(idx_all[big[:300]], out)
    assert np.array_equal(out, out_big[:300])
    m_small.close()
    # row-quad copies serve the exact part of such a pair (the default; `m` above was made without): the same taps, the same bits
    gpu_ctx.setQuadCopies("on")
    try:
        base_q = [E.RadonIntermediate.from_host(gpu_ctx, h, S, S) for h in host]
        mq = E.MetricRadonIntermediate(gpu_ctx, Ps, [base_q[v % 4] for v in range(n)]).setSmallEval(False).setRecordReuse(False).setSampling("polynomial")
    finally:
        gpu_ctx.setQuadCopies("auto")
    tq, vq = mq.evaluate_range(0, n_pairs, want_pairs=True)
    assert tq == total_p and np.array_equal(vq, vp)
    mq.close()
    # correlation variant: the moments of the polynomial part and of the exact part in one sum
    m.useCorrelation(True)
    oracle_mod.set_use_corr(1)
    try:
        want_c = oracle_mod.evaluate_pairs(Ps, dt_host, S, S, idx_all[sub])
    finally:
        oracle_mod.set_use_corr(0)
    out = np.empty(len(sub), np.float32)
    m.evaluate(idx_all[sub], out)
    assert np.abs(out - want_c["pairs"]).max() < 5e-5, (out, want_c["pairs"])
    cost = np.zeros((n, n), np.float32)
    m.evaluate(cost)  # the one-wave kernel with the cost image
    for q in sub:
        i, j = E.get_ij(int(q), n)
        assert abs(cost[j, i] - want_c["pairs"][list(sub).index(q)]) < 5e-5
    m.close()
    for d in base + base_q:
        d.close()


def test_auto_resolves_from_the_evaluation_not_the_shard(gpu_ctx):
    """ECC_SAMPLING_AUTO is a function of the size of the EVALUATION (n (n - 1) / 2 for evaluate_all and every range /
    shard of it): 40 views = 780 pairs -> polynomial, although every one of three shards has fewer than 512 pairs; a
    group of three ranks with the default mode therefore returns the rank-ordered sum of polynomial shard sums, and
    moving a shard boundary across 512 pairs changes nothing."""
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import synthetic
    rng = np.random.default_rng(5)
    n, S, B = 40, 128, 64
    Ps = synthetic.short_scan(n, S, S, 0.308 * 1024 / S)
    base = [E.RadonIntermediate.from_host(gpu_ctx, rng.standard_normal((B, B)).astype(np.float32), S, S) for _ in range(4)]
    dtrs = [base[v % 4] for v in range(n)]
    auto = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setSampling("auto")
    poly = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setSampling("polynomial")
    ref = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs).setSampling("reference")
    n_pairs = n * (n - 1) // 2
    assert n_pairs > 512
    for first, count in ((0, 260), (260, 260), (520, 260), (0, 511), (0, 513), (100, 1), (0, n_pairs)):
        a, va = auto.evaluate_range(first, count, want_pairs=True)
        b, vb = poly.evaluate_range(first, count, want_pairs=True)
        c, vc = ref.evaluate_range(first, count, want_pairs=True)
        assert a == b and np.array_equal(va, vb), (first, count)
        assert not np.array_equal(va, vc)  # the other arithmetic really is different at these sizes
    # index lists resolve from the list's length: 2 pairs -> the reference arithmetic
    idx = [(0, 5, 0, 5), (2, 30, 2, 30)]
    assert auto.evaluate(idx) == ref.evaluate(idx)
    # the single-process group with the library default (AUTO) on every rank
    saved = E.MetricRadonIntermediate.default_sampling
    E.MetricRadonIntermediate.default_sampling = None
    try:
        g = E.Group([0, 0, 0])
        gm = E.GroupMetricRadonIntermediate(g, Ps, dtrs)
        bnd = poly.balanced_shards(3)
        parts = [poly.evaluate_range(bnd[r], bnd[r + 1] - bnd[r]) for r in range(3)]
        assert min(bnd[r + 1] - bnd[r] for r in range(3)) < 512
        assert gm.evaluate() == (parts[0] + parts[1] + parts[2]) / n_pairs
        gm.close()
        g.close()
    finally:
        E.MetricRadonIntermediate.default_sampling = saved
    for m in (auto, poly, ref):
        m.close()
    for d in base:
        d.close()


def test_automatic_row_quad_copies_follow_the_scan(gpu_ctx):
    """ECC_QUAD_COPIES_AUTO (the default) decides once per metric, at its first evaluation of >= 32 768 pairs, from the matrices:
    a 200-degree short scan has pairs whose baseline passes through the object (kappa_max = pi/2: the only readers of the row-quad
    copies) -- built; a 60-degree scan has none -- 4x the stack's memory is not spent.  Small
    evaluations never build them.  The values are those of a metric without the copies either way."""
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import synthetic
    rng = np.random.default_rng(23)
    n, S, B = 260, 128, 32   # 33 670 pairs
    base = [E.RadonIntermediate.from_host(gpu_ctx, rng.standard_normal((B, B)).astype(np.float32), S, S) for _ in range(5)]
    dtrs = [base[v % 5] for v in range(n)]
    for span, expect in ((200.0, True), (60.0, False)):
        Ps = synthetic.short_scan(n, S, S, 0.308 * 1024 / S, span_deg=span)
        gpu_ctx.setQuadCopies("off")
        try:
            off = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs)
        finally:
            gpu_ctx.setQuadCopies("auto")
        m = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs)
        assert m.device_bytes()["quad_copies"] == 0            # nothing before the first large evaluation
        assert m.evaluate(set(range(40))) == off.evaluate(set(range(40)))
        assert m.device_bytes()["quad_copies"] == 0            # a 780-pair index list does not decide
        a, va = m.evaluate_range(0, n * (n - 1) // 2, want_pairs=True)
        b, vb = off.evaluate_range(0, n * (n - 1) // 2, want_pairs=True)
        assert a == b and np.array_equal(va, vb)
        q = m.device_bytes()["quad_copies"]
        assert (q == n * ((B + 1 + 3) // 4) * 64 * 64) if expect else (q == 0), (span, q)   # pitch 64 floats for 32 bins
        assert m.evaluate() == off.evaluate() and m.device_bytes()["quad_copies"] == q     # decided once
        assert off.device_bytes()["quad_copies"] == 0
        m.close(); off.close()
    small = E.MetricRadonIntermediate(gpu_ctx, synthetic.short_scan(60, S, S, 0.308 * 1024 / S), dtrs[:60])
    small.evaluate()
    assert small.device_bytes()["quad_copies"] == 0            # 1 770 pairs: never
    small.close()
    for d in base:
        d.close()
