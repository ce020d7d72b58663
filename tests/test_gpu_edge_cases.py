"""Edge cases of the hot path on the GPU (through the C ABI) against the oracle: smallest problems,
degenerate inputs, the non-derivative variant, error behaviour."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-30)


def test_two_and_three_views(gpu_ctx, oracle_mod, small_scan):
    import epipolarconsistency_amd as E
    s = small_scan
    for views in ([0, 5], [1, 4, 7]):
        Ps = [s["Ps"][v] for v in views]
        host = [s["dtrs"][v] for v in views]
        dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in host]
        m = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs)
        want = oracle_mod.evaluate_all(Ps, host, s["n_u"], s["n_v"])
        assert _rel(m.evaluate(), want["mean"]) < 1e-5


def test_single_view_is_rejected(gpu_ctx, small_scan):
    """The reference divides 0/0 for n < 2 (...RadonIntermediate.cpp:224); the ABI reports an error."""
    import epipolarconsistency_amd as E
    s = small_scan
    d = E.RadonIntermediate.from_host(gpu_ctx, s["dtrs"][0], s["n_u"], s["n_v"])
    m = E.MetricRadonIntermediate(gpu_ctx, s["Ps"][:1], [d])
    with pytest.raises(E.EccError):
        m.evaluate()


@pytest.mark.parametrize("shape,bins", [((8, 8), (5, 7)), ((3, 200), (64, 9)), ((130, 2), (7, 65))])
def test_tiny_and_ragged_radon(gpu_ctx, oracle_mod, shape, bins):
    import epipolarconsistency_amd as E
    rng = np.random.default_rng(2)
    img = rng.uniform(0, 9, size=shape).astype(np.float32)
    for filt in (0, 2):
        want = oracle_mod.radon(img, bins[0], bins[1], filter=filt)
        got = E.RadonIntermediate.compute(gpu_ctx, img, bins[0], bins[1], filter=filt).readback()
        assert np.array_equal(got, want)


def test_zero_image_and_constant_image(gpu_ctx, oracle_mod):
    import epipolarconsistency_amd as E
    z = np.zeros((40, 56), np.float32)
    assert np.all(E.RadonIntermediate.compute(gpu_ctx, z, 32, 24).readback() == 0)
    c = np.full((40, 56), 2.5, np.float32)
    assert np.array_equal(E.RadonIntermediate.compute(gpu_ctx, c, 32, 24).readback(), oracle_mod.radon(c, 32, 24))


def test_batch_equals_single(gpu_ctx, small_scan):
    import epipolarconsistency_amd as E
    s = small_scan
    batch = E.RadonIntermediate.compute_batch(gpu_ctx, s["imgs"][:3], 48, 40)
    for k in range(3):
        single = E.RadonIntermediate.compute(gpu_ctx, s["imgs"][k], 48, 40)
        assert np.array_equal(batch[k].readback(), single.readback())


def test_non_derivative_pairs(gpu_ctx, oracle_mod, small_scan):
    """Filter::None dtrs: the (alpha + pi, -t) fold keeps the sign (ref: ...RadonIntermediate.cu:80-83)."""
    import epipolarconsistency_amd as E
    s = small_scan
    host = [oracle_mod.radon(im, 64, 64, filter=2) for im in s["imgs"][:5]]
    dtrs = E.RadonIntermediate.compute_batch(gpu_ctx, s["imgs"][:5], 64, 64, filter=E.FILTER_NONE)
    assert not dtrs[0].isDerivative()
    for d, h in zip(dtrs, host):
        assert np.array_equal(d.readback(), h)
    m = E.MetricRadonIntermediate(gpu_ctx, s["Ps"][:5], dtrs)
    want = oracle_mod.evaluate_all(s["Ps"][:5], host, s["n_u"], s["n_v"], is_derivative=False)
    assert _rel(m.evaluate(), want["mean"]) < 1e-5


def test_identical_views_in_index_list(gpu_ctx, oracle_mod, small_scan):
    """(P, P) tuples hit the reference's same-pointer guard (EpipolarConsistencyCommon.hxx:108-113): value 0."""
    import epipolarconsistency_amd as E
    s = small_scan
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in s["dtrs"]]
    m = E.MetricRadonIntermediate(gpu_ctx, s["Ps"], dtrs)
    idx = np.array([[3, 3, 3, 3], [0, 1, 0, 1], [0, 1, 0, 1]], np.int32)
    out = np.full(3, -1, np.float32)
    m.evaluate(idx, out)
    want = oracle_mod.evaluate_pairs(s["Ps"], s["dtrs"], s["n_u"], s["n_v"], idx)
    assert out[0] == 0.0 and want["pairs"][0] == 0.0
    assert out[1] == out[2] and _rel(out[1], want["pairs"][1]) < 2e-4


def test_wide_kappa_range_general_branch(gpu_ctx, oracle_mod, small_scan):
    """A huge object radius makes kappa_max = pi/2: epipolar lines of every orientation, so the general
    (non-steep) branches of the angle computation and the texture clamp are exercised."""
    import epipolarconsistency_amd as E
    s = small_scan
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in s["dtrs"]]
    m = E.MetricRadonIntermediate(gpu_ctx, s["Ps"], dtrs)
    m.setObjectRadius(5000.0)
    want = oracle_mod.evaluate_all(s["Ps"], s["dtrs"], s["n_u"], s["n_v"], object_radius_mm=5000.0)
    total, vals = m.evaluate_range(0, 28, want_pairs=True)
    assert _rel(total / 28, want["mean"]) < 1e-5
    np.testing.assert_allclose(vals, want["pairs"], rtol=2e-4)


def test_rotated_detector_geometry(gpu_ctx, oracle_mod, small_scan):
    """In-plane detector rotations of 30..150 deg: epipolar lines far from horizontal."""
    import epipolarconsistency_amd as E
    s = small_scan
    Ps = []
    for k, P in enumerate(s["Ps"]):
        a = np.deg2rad(30.0 + 17.0 * k)
        c, sn = np.cos(a), np.sin(a)
        T = np.array([[1, 0, 64.0], [0, 1, 64.0], [0, 0, 1]]) @ np.array([[c, -sn, 0], [sn, c, 0], [0, 0, 1]]) @ \
            np.array([[1, 0, -64.0], [0, 1, -64.0], [0, 0, 1]])
        Ps.append(T @ P)
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in s["dtrs"]]
    m = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs)
    want = oracle_mod.evaluate_all(Ps, s["dtrs"], s["n_u"], s["n_v"])
    total, vals = m.evaluate_range(0, 28, want_pairs=True)
    assert _rel(total / 28, want["mean"]) < 1e-5
    np.testing.assert_allclose(vals, want["pairs"], rtol=2e-4)


def test_error_behaviour(gpu_ctx, small_scan):
    import epipolarconsistency_amd as E
    s = small_scan
    img = s["imgs"][0]
    with pytest.raises(E.EccError) as ei:
        E.RadonIntermediate.compute(gpu_ctx, img, 32, 32, filter=7)  # unknown filter
    assert ei.value.code == 1
    with pytest.raises(E.EccError):
        E.RadonIntermediate.compute(gpu_ctx, img, 0, 32)
    a = E.RadonIntermediate.compute(gpu_ctx, img, 32, 32)
    b = E.RadonIntermediate.compute(gpu_ctx, img, 40, 32)
    with pytest.raises(E.EccError):
        E.MetricRadonIntermediate(gpu_ctx, s["Ps"][:2], [a, b])  # mixed bin counts are rejected
    m = E.MetricRadonIntermediate(gpu_ctx, None, [a, a])
    with pytest.raises(E.EccError):
        m.evaluate()  # projection matrices not set
    m.setProjectionMatrices(s["Ps"][:2])
    assert np.isfinite(m.evaluate())
    with pytest.raises(E.EccError):
        m.evaluate_range(0, 2)  # only one pair exists
    m3 = E.MetricRadonIntermediate(gpu_ctx, s["Ps"][:3], [a, a])
    with pytest.raises(E.EccError):
        m3.evaluate()  # fewer dtrs than projection matrices


def test_reevaluate_after_changing_one_view(gpu_ctx, oracle_mod, small_scan):
    """The optimiser loop (ref: Gui/SingleImageMotion.h:84-90): overwrite one matrix, set all, evaluate."""
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import geometry
    s = small_scan
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in s["dtrs"]]
    m = E.MetricRadonIntermediate(gpu_ctx, s["Ps"], dtrs)
    packed = E.pack_projection_matrices(s["Ps"])
    for step in range(3):
        T = geometry.rigid_transform(tx=0.7 * step, ry=0.01 * step)
        Ps = list(s["Ps"])
        Ps[3] = s["Ps"][3] @ T
        packed[3] = Ps[3].T.reshape(12)
        m.setProjectionMatrices(packed)
        want = oracle_mod.evaluate_all(Ps, s["dtrs"], s["n_u"], s["n_v"])
        assert _rel(m.evaluate(), want["mean"]) < 1e-5


def test_dtr_save_load(gpu_ctx, small_scan, tmp_path):
    import os
    import epipolarconsistency_amd as E
    s = small_scan
    d = E.RadonIntermediate.compute(gpu_ctx, s["imgs"][2], 48, 40)
    p = os.path.join(tmp_path, "d.nrrd")
    d.save(p, projection_matrix=s["Ps"][2])
    d2, info = E.RadonIntermediate.load(gpu_ctx, p)
    assert np.array_equal(d2.readback(), d.readback()) and d2.isDerivative()
    assert d2.getOriginalImageSize(0) == s["n_u"] and d2.getRadonBinNumber(1) == 40
    assert np.allclose(info["projection_matrix"], s["Ps"][2], rtol=1e-11)


def test_use_correlation(gpu_ctx, oracle_mod, small_scan):
    """useCorrelation(true): 1 - cc per pair (provisional formula, parity unpinned).  1 - cc is
    a cancellation of two numbers near 1, so it is compared absolutely."""
    import epipolarconsistency_amd as E
    s = small_scan
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in s["dtrs"]]
    m = E.MetricRadonIntermediate(gpu_ctx, s["Ps"], dtrs)
    m.useCorrelation(True)
    oracle_mod.set_use_corr(1)
    try:
        want = oracle_mod.evaluate_all(s["Ps"], s["dtrs"], s["n_u"], s["n_v"])
        idx = np.array([[0, 4, 0, 4], [6, 2, 6, 2]], np.int32)
        want_idx = oracle_mod.evaluate_pairs(s["Ps"], s["dtrs"], s["n_u"], s["n_v"], idx)
    finally:
        oracle_mod.set_use_corr(0)
    cost = np.zeros((8, 8), np.float32)
    mean = m.evaluate(cost)
    assert abs(mean - want["mean"]) < 2e-6 and 0 < mean < 0.1
    for ij in range(28):
        i, j = E.get_ij(ij, 8)
        assert abs(cost[j, i] - want["pairs"][ij]) < 2e-6
    out = np.zeros(2, np.float32)
    assert abs(m.evaluate(idx, out) - want_idx["mean"]) < 2e-6
    m.useCorrelation(False)
    ssd = oracle_mod.evaluate_all(s["Ps"], s["dtrs"], s["n_u"], s["n_v"])
    assert abs(m.evaluate() - ssd["mean"]) < 1e-5 * ssd["mean"]


@pytest.mark.parametrize("shape,bins", [((96, 128), (96, 80)), ((61, 47), (33, 29)), ((40, 40), (8, 300)),
                                        ((40, 40), (6, 1000)),    # more than 768 distance bins: four outputs per thread
                                        ((24, 24), (3, 3400))])   # table + row above 64 KB of LDS: table from global memory
def test_ramp_filtered_radon(gpu_ctx, oracle_mod, shape, bins):
    """Filter::Ramp: line integrals + ramp filter along t (ref: RadonIntermediate.cu:166-167,173-237) --
    same binary64 circular convolution as the oracle, rounded once."""
    import epipolarconsistency_amd as E
    rng = np.random.default_rng(11)
    img = rng.uniform(0, 4, size=shape).astype(np.float32)
    want = oracle_mod.radon(img, bins[0], bins[1], filter=1)
    d = E.RadonIntermediate.compute(gpu_ctx, img, bins[0], bins[1], filter=E.FILTER_RAMP)
    assert d.getFilter() == E.FILTER_RAMP and not d.isDerivative()
    got = d.readback()
    assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max()
    # the filter removes the mean of every angle column
    assert np.abs(got.astype(np.float64).sum(axis=0)).max() <= 1e-4 * np.abs(got).sum(axis=0).max()


def test_ramp_filtered_pairs(gpu_ctx, oracle_mod, small_scan):
    """Ramp-filtered dtrs go through the non-derivative pair kernel (no sign flip on the fold)."""
    import epipolarconsistency_amd as E
    s = small_scan
    host = [oracle_mod.radon(im, 64, 64, filter=1) for im in s["imgs"][:4]]
    dtrs = E.RadonIntermediate.compute_batch(gpu_ctx, s["imgs"][:4], 64, 64, filter=E.FILTER_RAMP)
    for d, h in zip(dtrs, host):
        assert np.abs(d.readback() - h).max() <= 1e-6 * np.abs(h).max()
    m = E.MetricRadonIntermediate(gpu_ctx, s["Ps"][:4], dtrs)
    want = oracle_mod.evaluate_all(s["Ps"][:4], host, s["n_u"], s["n_v"], is_derivative=False)
    assert _rel(m.evaluate(), want["mean"]) < 1e-5


@pytest.mark.parametrize("dkappa", [0.0, 0.004])
def test_evaluate_for_image_pair(gpu_ctx, oracle_mod, small_scan, dkappa):
    """E7 (ref: ...RadonIntermediate.cpp:324-393): redundant signals of one pair, GPU vs the oracle's restatement
    of the evident intent."""
    import epipolarconsistency_amd as E
    s = small_scan
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in s["dtrs"]]
    m = E.MetricRadonIntermediate(gpu_ctx, s["Ps"], dtrs).setEpipolarPlaneStep(dkappa)
    for (i, j) in ((1, 5), (6, 2)):
        ecc, got = m.evaluateForImagePair(i, j)
        want = oracle_mod.evaluate_for_image_pair(s["Ps"], s["dtrs"], i, j, s["n_u"], s["n_v"], dkappa=dkappa)
        n = len(want["kappas"])
        assert len(got["kappas"]) == n and n > 50
        assert np.array_equal(got["kappas"], want["kappas"])
        assert np.allclose(got["K01"], want["K01"], rtol=3e-7, atol=1e-7 * np.abs(want["K01"]).max())
        assert np.abs(got["radon_samples0"] - want["radon0"]).max() < 2e-6
        assert np.abs(got["radon_samples1"] - want["radon1"]).max() < 2e-6
        scale = max(np.abs(want["samples0"]).max(), np.abs(want["samples1"]).max())
        assert np.abs(got["redundant_samples0"] - want["samples0"]).max() < 1e-3 * scale
        assert np.abs(got["redundant_samples1"] - want["samples1"]).max() < 1e-3 * scale
        assert _rel(ecc, want["ecc"]) < 1e-4
    # the two signals agree much better for the consistent geometry than for a perturbed one
    ecc_ok, _ = m.evaluateForImagePair(1, 5)
    Ps_bad = [p.copy() for p in s["Ps"]]
    Ps_bad[5] = Ps_bad[5] @ E.geometry.rigid_transform(tx=3.0, ry=0.02)
    m.setProjectionMatrices(Ps_bad)
    ecc_bad, _ = m.evaluateForImagePair(1, 5)
    assert ecc_bad > 2 * ecc_ok
    with pytest.raises(E.EccError):
        m.evaluateForImagePair(0, 99)


def test_host_sampling_helpers_and_replace(gpu_ctx, oracle_mod):
    """RadonIntermediate::tex2D / sample (host, RadonIntermediate.h:86-108) and replaceRadonIntermediateData
    (RadonIntermediate.cpp:105-123)."""
    import epipolarconsistency_amd as E
    rng = np.random.default_rng(2)
    n_t, n_alpha, n_u, n_v = 40, 56, 128, 96
    a = rng.standard_normal((n_t, n_alpha)).astype(np.float32)
    d = E.RadonIntermediate.from_host(gpu_ctx, a, n_u, n_v)
    with pytest.raises(ValueError):
        d.tex2D(0.5, 0.5)
    assert np.array_equal(d.readback(), a) and d.data() is not None
    # grid points of the (n - 1)-scaled lattice return the texels themselves, the far corner is clamped
    for (i, j) in ((0, 0), (7, 3), (n_alpha - 1, n_t - 1), (20, n_t - 1)):
        assert d.tex2D(i / (n_alpha - 1), j / (n_t - 1)) == pytest.approx(float(a[j, i]), rel=1e-6, abs=1e-6)
    mid = d.tex2D(0.5 / (n_alpha - 1), 0.0)
    assert mid == pytest.approx(0.5 * (float(a[0, 0]) + float(a[0, 1])), rel=1e-6)
    # sample(line): location from lineToSampleDtr, sign flipped on the folded branch of a derivative dtr
    range_t = np.float32(d.getRadonBinSize(1)) * np.float32(n_t)
    for line in ([0.6, 0.8, 12.0], [0.6, -0.8, -30.0], [-1.0, 0.1, 5.0]):
        l = np.array(line, np.float32)
        want_loc, folded = oracle_mod.line_to_sample_dtr(l, float(range_t))
        got = d.sample(l)
        assert np.array_equal(l[:2], want_loc[:2])
        assert got == (-1.0 if folded else 1.0) * d.tex2D(want_loc[0], want_loc[1])
    # new data: the handle is replaced, metadata kept, bin size follows the new shape
    b = rng.standard_normal((64, 48)).astype(np.float32)
    d.replaceRadonIntermediateData(b)
    assert d.getRadonBinNumber(0) == 48 and d.getRadonBinNumber(1) == 64 and d.getOriginalImageSize(0) == n_u
    assert d.isDerivative() and np.array_equal(d.readback(), b)
    assert abs(d.getRadonBinSize(1) - np.sqrt(n_u ** 2 + n_v ** 2) / 64) < 1e-9
    d.clearRawData()
    assert d.data() is None


def test_large_radon_intermediates_1448_bins(gpu_ctx, oracle_mod):
    """Radon intermediates of 1448 x 1448 bins (row-paired copy 17 MB > 2^24 bytes): the pair kernel switches to integer
    offset arithmetic (EccPairParams::wide_offsets); the reference has no size limit (RadonIntermediate.cpp:198-211).
    Both throughput paths and the reference-arithmetic path against the oracle."""
    import epipolarconsistency_amd as E
    from conftest import make_small_scan
    Ps, imgs = make_small_scan(n=5)
    B = 1448
    dtrs = E.RadonIntermediate.compute_batch(gpu_ctx, imgs, B, B)
    host = [d.readback() for d in dtrs]
    bins = np.random.default_rng(4).integers(0, B * B, size=2000).astype(np.int32)
    assert np.array_equal(host[3].reshape(-1)[bins], oracle_mod.radon_bins(imgs[3], B, B, bins))
    want = oracle_mod.evaluate_all(Ps, host, 128, 128)
    m = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs)
    for mode, pair_tol in (("polynomial", 5e-4), ("per_sample", 5e-4), ("reference", 1e-6)):
        total, vals = m.setSampling(mode).evaluate_range(0, 10, want_pairs=True)
        assert abs(total / 10 - want["mean"]) <= (1e-6 if mode == "reference" else 5e-5) * want["mean"], mode
        np.testing.assert_allclose(vals, want["pairs"], rtol=pair_tol, err_msg=mode)
    m.close()


def test_refresh_after_recomputing_slabs_in_place(gpu_ctx, oracle_mod, small_scan):
    """A metric samples a snapshot (row-paired copies) of its dtrs: slabs recomputed in place are seen after
    refreshRadonIntermediates() (ecc_metric_refresh_dtrs), and only then -- except in reference mode, which reads the
    slabs themselves."""
    import torch
    import epipolarconsistency_amd as E
    s = small_scan
    dev = torch.device("cuda", gpu_ctx.device)
    imgs = torch.from_numpy(np.ascontiguousarray(s["imgs"])).to(dev)
    slabs = torch.zeros((8, E.slab_floats(96, 96)), dtype=torch.float32, device=dev)
    dtrs = E.RadonIntermediate.compute_into(gpu_ctx, imgs, slabs, 96, 96)
    m = E.MetricRadonIntermediate(gpu_ctx, s["Ps"], dtrs).setSampling("polynomial")
    before = m.evaluate()
    # new image content for views 2 and 5, same slabs
    imgs2 = imgs.clone()
    imgs2[2] *= 1.5
    imgs2[5] = torch.flip(imgs2[5], dims=[1])
    keep = E.RadonIntermediate.compute_into(gpu_ctx, imgs2[2:3], slabs[2:3], 96, 96)
    keep += E.RadonIntermediate.compute_into(gpu_ctx, imgs2[5:6], slabs[5:6], 96, 96)
    gpu_ctx.synchronize()
    assert m.evaluate() == before  # still the snapshot
    host = [oracle_mod.radon(im, 96, 96) for im in imgs2.cpu().numpy()]
    want = oracle_mod.evaluate_all(s["Ps"], host, 128, 128)["mean"]
    assert abs(m.setSampling("reference").evaluate() - want) <= 1e-6 * want  # reads the live slabs
    m.setSampling("polynomial")
    m.refreshRadonIntermediates(2, 1)
    m.refreshRadonIntermediates(5, 1)
    after = m.evaluate()
    assert abs(after - want) <= 1e-5 * want and abs(after - before) > 1e-3 * before
    with pytest.raises(E.EccError):
        m.refreshRadonIntermediates(7, 2)
    m.close()
