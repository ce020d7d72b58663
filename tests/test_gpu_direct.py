"""MetricDirect on the device against the oracle's restatement of computeForImagePair
(ref: EpipolarConsistencyDirect.cpp:67-219, EpipolarConsistencyDirect.cu:31-125)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-30)


@pytest.mark.parametrize("dkappa", [0.0, 0.002])
def test_pair_signals_and_metric(gpu_ctx, oracle_mod, small_scan, dkappa):
    import epipolarconsistency_amd as E
    s = small_scan
    imgs = np.ascontiguousarray(s["imgs"], np.float32)
    m = E.MetricDirect(gpu_ctx, s["Ps"], imgs).setEpipolarPlaneStep(dkappa)
    radius = oracle_mod.object_radius(s["Ps"][0], s["n_u"], s["n_v"])
    assert abs(m.getObjectRadius() - radius) < 1e-9
    for (i, j) in ((1, 5), (6, 2), (0, 7)):
        val, got = m.evaluateForImagePair(i, j)
        want = oracle_mod.direct_pair(s["Ps"][i], s["Ps"][j], imgs[i], imgs[j], dkappa, radius)
        n = len(want["kappas"])
        assert len(got["kappas"]) == n and n > 100
        assert np.array_equal(got["kappas"], want["kappas"])
        # lines: float64 geometry rounded to float on both sides (sin/cos of two libms)
        assert np.abs(got["lines"] - want["lines"]).max() <= 2e-6 * np.abs(want["lines"]).max()
        scale = max(np.abs(want["samples0"]).max(), np.abs(want["samples1"]).max())
        same = np.all(got["lines"] == want["lines"], axis=1)
        assert same.mean() > 0.9
        # identical lines give bit-identical line integrals
        assert np.array_equal(got["redundant_samples0"][same], want["samples0"][same])
        assert np.array_equal(got["redundant_samples1"][same], want["samples1"][same])
        assert np.abs(got["redundant_samples0"] - want["samples0"]).max() <= 2e-3 * scale
        assert _rel(val, want["metric"]) < 1e-5


def test_all_pairs_sum_and_cost_image(gpu_ctx, oracle_mod, small_scan):
    import torch
    import epipolarconsistency_amd as E
    s = small_scan
    n = 5
    imgs = np.ascontiguousarray(s["imgs"][:n], np.float32)
    Ps = s["Ps"][:n]
    want = oracle_mod.direct_evaluate(Ps, imgs)
    m = E.MetricDirect(gpu_ctx, Ps, torch.from_numpy(imgs).cuda())  # borrowed device images
    cost = np.full((n, n), -1.0, np.float32)
    got = m.evaluate(cost)
    assert _rel(got, want["sum"]) < 1e-5
    iu = np.triu_indices(n, 1)
    assert np.allclose(cost.T[iu], want["cost"].T[iu], rtol=1e-5)
    assert np.all(cost[iu] == -1.0) and np.all(np.diag(cost) == -1.0)  # untouched entries survive
    assert _rel(m.evaluate(), want["sum"]) < 1e-5
    # consistent geometry scores far better than a perturbed one
    Pb = [p.copy() for p in Ps]
    Pb[2] = Pb[2] @ E.geometry.rigid_transform(tx=4.0, rz=0.03)
    assert m.setProjectionMatrices(Pb).evaluate() > 1.05 * got
    # user radius
    r = oracle_mod.direct_evaluate(Ps, imgs, object_radius_mm=40.0)
    assert _rel(m.setProjectionMatrices(Ps).setObjectRadius(40.0).evaluate(), r["sum"]) < 1e-5


def test_non_square_images_and_errors(gpu_ctx, oracle_mod):
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import synthetic
    n_u, n_v = 96, 64
    Ps = synthetic.short_scan(3, n_u, n_v, 3.0)
    imgs = synthetic.projections_numpy(Ps, n_u, n_v, synthetic.sphere_phantom(extent_mm=25, rmin=6, rmax=18))
    m = E.MetricDirect(gpu_ctx, Ps, imgs)
    val, got = m.evaluateForImagePair(0, 2)
    want = oracle_mod.direct_pair(Ps[0], Ps[2], imgs[0], imgs[2], 0.0, oracle_mod.object_radius(Ps[0], n_u, n_v))
    assert len(got["kappas"]) == len(want["kappas"]) and _rel(val, want["metric"]) < 1e-5
    with pytest.raises(E.EccError):
        m.evaluateForImagePair(0, 3)
    m2 = E.MetricDirect(gpu_ctx, None, imgs)
    with pytest.raises(E.EccError):
        m2.evaluate()


def test_fan_beam_consistency_variant(gpu_ctx, oracle_mod, small_scan):
    """setFanBeamConsistency (ref: RectifiedFBCC.h, EpipolarConsistencyDirect.cpp:133-196, .cu:87-101)."""
    import epipolarconsistency_amd as E
    s = small_scan
    imgs = np.ascontiguousarray(s["imgs"][:4], np.float32)
    Ps = s["Ps"][:4]
    radius = oracle_mod.object_radius(Ps[0], s["n_u"], s["n_v"])
    m = E.MetricDirect(gpu_ctx, Ps, imgs).setFanBeamConsistency(True)
    for (i, j) in ((0, 3), (2, 1)):
        val, got = m.evaluateForImagePair(i, j)
        want = oracle_mod.direct_pair(Ps[i], Ps[j], imgs[i], imgs[j], 0.0, radius, fbcc=True)
        assert len(got["kappas"]) == len(want["kappas"])
        scale = np.abs(want["samples0"]).max()
        assert np.abs(got["redundant_samples0"] - want["samples0"]).max() <= 1e-5 * scale
        assert np.abs(got["redundant_samples1"] - want["samples1"]).max() <= 1e-5 * scale
        # the fan-beam condition: the two weighted signals agree for consistent data
        assert np.corrcoef(got["redundant_samples0"], got["redundant_samples1"])[0, 1] > 0.999
        # (measured: identical lines give bit-identical weighted integrals; 200 random configurations of
        # scripts/fuzz_direct.py agree to 1e-15 -- the bar leaves room for a line that differs by one float ulp)
        assert _rel(val, want["metric"]) < 1e-5
    want = oracle_mod.direct_evaluate(Ps, imgs, fbcc=True)
    assert _rel(m.evaluate(), want["sum"]) < 1e-5
    # switching back gives the derivative form again
    assert _rel(m.setFanBeamConsistency(False).evaluate(), oracle_mod.direct_evaluate(Ps, imgs)["sum"]) < 1e-5


def test_caller_provided_kappa_grid(gpu_ctx, small_scan):
    """ref: computeForImagePair takes a non-empty `kappas` vector as the grid (EpipolarConsistencyDirect.cpp:105-117).
    The automatic grid handed back in reproduces the same bits; a thinned grid picks out exactly those lines and the
    metric is the sum over them with the same dkappa."""
    import epipolarconsistency_amd as E
    s = small_scan
    d = E.MetricDirect(gpu_ctx, s["Ps"][:3], np.ascontiguousarray(s["imgs"][:3]))
    m0, a = d.evaluateForImagePair(0, 2)
    m1, b = d.evaluateForImagePair(0, 2, kappas=a["kappas"])
    assert m1 == m0 and np.array_equal(b["redundant_samples0"], a["redundant_samples0"])
    assert np.array_equal(b["redundant_samples1"], a["redundant_samples1"]) and np.array_equal(b["lines"], a["lines"])
    thin = a["kappas"][::3]
    m2, c = d.evaluateForImagePair(0, 2, kappas=thin)
    assert np.array_equal(c["redundant_samples0"], a["redundant_samples0"][::3])
    assert np.array_equal(c["redundant_samples1"], a["redundant_samples1"][::3])
    dk = float(a["kappas"][1] - a["kappas"][0])
    diff = (c["redundant_samples0"] - c["redundant_samples1"]).astype(np.float32)
    want = float(np.sum((diff * diff).astype(np.float64)) * (m0 / np.sum(((a["redundant_samples0"] - a["redundant_samples1"]).astype(np.float32) ** 2).astype(np.float64))))
    assert abs(m2 - want) <= 1e-9 * abs(want) and dk > 0
    with pytest.raises(E.EccError):
        d.evaluateForImagePair(0, 2, kappas=np.zeros(0, np.float32))
    d.close()


@pytest.mark.parametrize("fbcc", [False, True])
def test_larger_images_take_the_lds_slab_kernel(gpu_ctx, oracle_mod, fbcc):
    """From 384 pixels per side on the line integrals go through the LDS slab tile the Radon kernel uses
    (csrc/ecc_slab_tile.h); smaller images (everything above) gather from global memory.  Same arithmetic: identical
    lines give bit-identical integrals, the metric agrees to 1e-5, in both forms and for a non-square image whose lines
    run closer to x (transposed tile) as well as to y."""
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import geometry, synthetic
    n_u, n_v = 448, 400
    Ps = synthetic.short_scan(4, n_u, n_v, 0.308 * 1024 / n_u)
    Ps[2] = Ps[2] @ geometry.rigid_transform(tz=40.0, rx=0.5)  # a pair whose epipolar lines are steep in the image
    imgs = np.ascontiguousarray(synthetic.projections_numpy(Ps, n_u, n_v, synthetic.sphere_phantom()), np.float32)
    radius = oracle_mod.object_radius(Ps[0], n_u, n_v)
    m = E.MetricDirect(gpu_ctx, Ps, imgs).setFanBeamConsistency(fbcc)
    for (i, j) in ((0, 3), (2, 1), (0, 2)):
        val, got = m.evaluateForImagePair(i, j)
        want = oracle_mod.direct_pair(Ps[i], Ps[j], imgs[i], imgs[j], 0.0, radius, fbcc=fbcc)
        assert len(got["kappas"]) == len(want["kappas"]) > 1000
        same = np.all(got["lines"] == want["lines"], axis=1)
        assert same.mean() > 0.9
        assert np.array_equal(got["redundant_samples0"][same], want["samples0"][same])
        assert np.array_equal(got["redundant_samples1"][same], want["samples1"][same])
        assert _rel(val, want["metric"]) < 1e-5
    assert _rel(m.evaluate(), oracle_mod.direct_evaluate(Ps, imgs, fbcc=fbcc)["sum"]) < 1e-5
    m.close()
