"""The oracle reproduces the committed golden fixtures (tests/golden/*.npz, made by
tests/golden/make_golden.py) exactly; on the GPU box the HIP path is held to the same fixtures."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _checksum(a):
    a = np.ascontiguousarray(a, np.float32)
    return np.array([a.astype(np.float64).sum(), np.abs(a).astype(np.float64).sum(),
                     float(np.bitwise_xor.reduce(a.view(np.uint32).reshape(-1)))])


def test_oracle_reproduces_example_pair(oracle_mod):
    g = np.load(os.path.join(G, "example_pair_256.npz"))
    imgs, Ps = g["images"], list(g["Ps"])
    dtrs = [oracle_mod.radon(im, int(g["n_alpha"]), int(g["n_t"])) for im in imgs]
    for k, d in enumerate(dtrs):
        assert np.array_equal(_checksum(d), g["dtr_checksums"][k])
        assert np.array_equal(d.reshape(-1)[g["sample_bins"]], g["dtr_samples"][k])
    res = oracle_mod.evaluate_all(Ps, dtrs, 256, 190, want_K01=True)
    assert res["pairs"][0] == g["pair_value"] and res["n_kappa"] == int(g["n_kappa"])
    assert np.array_equal(res["K01s"][0], g["K01"])


def test_oracle_reproduces_synthetic8(oracle_mod, small_scan):
    g = np.load(os.path.join(G, "synthetic8_128.npz"))
    for k in range(8):
        assert np.array_equal(_checksum(small_scan["imgs"][k]), g["image_checksums"][k]), "generator drifted"
        assert np.array_equal(_checksum(small_scan["dtrs"][k]), g["dtr_checksums"][k])
    res = oracle_mod.evaluate_all(small_scan["Ps"], small_scan["dtrs"], 128, 128)
    assert np.array_equal(res["pairs"], g["pairs"]) and res["mean"] == float(g["mean"])


@pytest.mark.gpu
def test_hip_matches_example_pair_golden(gpu_ctx):
    import epipolarconsistency_amd as E
    g = np.load(os.path.join(G, "example_pair_256.npz"))
    dtrs = E.RadonIntermediate.compute_batch(gpu_ctx, g["images"], int(g["n_alpha"]), int(g["n_t"]))
    for k, d in enumerate(dtrs):
        got = d.readback()
        assert np.array_equal(_checksum(got), g["dtr_checksums"][k])
        assert np.array_equal(got.reshape(-1)[g["sample_bins"]], g["dtr_samples"][k])
    m = E.MetricRadonIntermediate(gpu_ctx, list(g["Ps"]), dtrs)
    assert abs(m.getObjectRadius() - float(g["object_radius"])) < 1e-9
    cost = np.zeros((2, 2), np.float32)
    mean = m.evaluate(cost)
    assert abs(mean - float(g["mean"])) <= 1e-5 * float(g["mean"])
    assert abs(cost[1, 0] - float(g["pair_value"])) <= 1e-5 * float(g["pair_value"])
    np.testing.assert_allclose(m.debug_K01(0, 1)[0], g["K01"], rtol=3e-6, atol=1e-9)


@pytest.mark.gpu
def test_hip_matches_synthetic8_golden(gpu_ctx, small_scan):
    import epipolarconsistency_amd as E
    g = np.load(os.path.join(G, "synthetic8_128.npz"))
    dtrs = E.RadonIntermediate.compute_batch(gpu_ctx, small_scan["imgs"], 96, 96)
    for k, d in enumerate(dtrs):
        assert np.array_equal(_checksum(d.readback()), g["dtr_checksums"][k])
    m = E.MetricRadonIntermediate(gpu_ctx, small_scan["Ps"], dtrs)
    total, vals = m.evaluate_range(0, 28, want_pairs=True)
    assert abs(total / 28 - float(g["mean"])) <= 1e-5 * float(g["mean"])
    np.testing.assert_allclose(vals, g["pairs"], rtol=2e-4)


def _rows_inputs():
    g = np.load(os.path.join(G, "example_pair_256.npz"))
    r = np.load(os.path.join(G, "example_pair_256_rows.npz"))
    return g, r, g["images"], list(g["Ps"]), int(g["n_alpha"]), int(g["n_t"])


def test_oracle_reproduces_widened_rows(oracle_mod):
    g, r, imgs, Ps, n_alpha, n_t = _rows_inputs()
    ramp = oracle_mod.radon(imgs[0], n_alpha, n_t, filter=1)
    assert np.array_equal(_checksum(ramp), r["ramp_checksum"]) and np.array_equal(ramp.reshape(-1)[g["sample_bins"]], r["ramp_samples"])
    pre = oracle_mod.preprocess(imgs[1] + 1.0, Ps[1], apply_log=True, scale=0.01)
    assert np.array_equal(_checksum(pre), r["pre_checksum"]) and np.array_equal(pre.reshape(-1)[r["pre_pixels"]], r["pre_samples"])
    dtrs = [oracle_mod.radon(im, n_alpha, n_t) for im in imgs]
    e7 = oracle_mod.evaluate_for_image_pair(Ps, dtrs, 0, 1, 256, 190)
    assert e7["ecc"] == float(r["e7_ecc"]) and len(e7["kappas"]) == int(r["e7_n"])
    assert np.array_equal(e7["samples0"][::8], r["e7_samples0"]) and np.array_equal(e7["samples1"][::8], r["e7_samples1"])
    oracle_mod.set_use_corr(True)
    try:
        assert oracle_mod.evaluate_all(Ps, dtrs, 256, 190)["pairs"][0] == r["corr_value"]
    finally:
        oracle_mod.set_use_corr(False)
    radius = oracle_mod.object_radius(Ps[0], 256, 190)
    d = oracle_mod.direct_pair(Ps[0], Ps[1], imgs[0], imgs[1], 0.0, radius)
    assert d["metric"] == float(r["direct_metric"]) and len(d["kappas"]) == int(r["direct_n"])
    assert np.array_equal(d["samples0"][::16], r["direct_samples0"]) and np.array_equal(d["samples1"][::16], r["direct_samples1"])
    f = oracle_mod.direct_pair(Ps[0], Ps[1], imgs[0], imgs[1], 0.0, radius, fbcc=True)
    assert f["metric"] == float(r["fbcc_metric"]) and np.array_equal(f["samples0"][::16], r["fbcc_samples0"])


@pytest.mark.gpu
def test_hip_matches_widened_rows_golden(gpu_ctx):
    import epipolarconsistency_amd as E
    g, r, imgs, Ps, n_alpha, n_t = _rows_inputs()
    ramp = E.RadonIntermediate.compute(gpu_ctx, imgs[0], n_alpha, n_t, filter=E.FILTER_RAMP).readback()
    assert np.abs(ramp.reshape(-1)[g["sample_bins"]] - r["ramp_samples"]).max() <= 1e-6 * np.abs(r["ramp_samples"]).max()
    pp = E.PreProccess()
    pp.intensity.apply_log, pp.intensity.scale = True, 0.01
    pre = pp.process(gpu_ctx, (imgs[1] + 1.0)[None], [Ps[1]])[0]
    assert np.abs(pre.reshape(-1)[r["pre_pixels"]] - r["pre_samples"]).max() <= 2e-7 * np.abs(r["pre_samples"]).max()
    dtrs = E.RadonIntermediate.compute_batch(gpu_ctx, imgs, n_alpha, n_t)
    m = E.MetricRadonIntermediate(gpu_ctx, Ps, dtrs)
    ecc, s = m.evaluateForImagePair(0, 1)
    assert len(s["kappas"]) == int(r["e7_n"]) and abs(ecc - float(r["e7_ecc"])) <= 1e-4 * float(r["e7_ecc"])
    corr = np.zeros(1, np.float32)
    m.useCorrelation(True).evaluate(np.array([[0, 1, 0, 1]], np.int32), corr)
    assert abs(corr[0] - float(r["corr_value"])) <= 2e-4 * abs(float(r["corr_value"])) + 1e-7
    d = E.MetricDirect(gpu_ctx, Ps, imgs)
    val, sd = d.evaluateForImagePair(0, 1)
    assert len(sd["kappas"]) == int(r["direct_n"]) and abs(val - float(r["direct_metric"])) <= 1e-5 * float(r["direct_metric"])
    val, _ = d.setFanBeamConsistency(True).evaluateForImagePair(0, 1)
    assert abs(val - float(r["fbcc_metric"])) <= 1e-3 * float(r["fbcc_metric"])


_RADON_VARIANTS = dict(deriv=(0, 0), deriv_sqrt=(0, 1), deriv_log=(0, 2), plain=(2, 0), ramp=(1, 0))


def test_oracle_reproduces_variants(oracle_mod, small_scan):
    g = np.load(os.path.join(G, "variants_128.npz"))
    for name, (f, post) in _RADON_VARIANTS.items():
        d = oracle_mod.radon(g["image"], 96, 80, filter=f, post=post)
        assert np.array_equal(_checksum(d), g["radon_%s_checksum" % name]), name
        assert np.array_equal(d.reshape(-1)[g["bins"]], g["radon_%s_samples" % name]), name
    s = small_scan
    r = oracle_mod.evaluate_pairs(s["Ps"], s["dtrs"], 128, 128, g["idx"])
    assert np.array_equal(r["pairs"], g["idx_pairs"]) and r["mean"] == float(g["idx_mean"])
    r = oracle_mod.evaluate_all(s["Ps"], s["dtrs"], 128, 128, object_radius_mm=25.0, dkappa=0.004)
    assert np.array_equal(r["pairs"], g["param_pairs"]) and r["n_kappa"] == int(g["param_n_kappa"])


@pytest.mark.gpu
def test_hip_matches_variants_golden(gpu_ctx, small_scan):
    """Every Radon filter / post-process bit-exact against the fixture (Ramp: 1e-6 of the maximum); index-list,
    subset and user-parameter evaluations against the fixture's oracle values."""
    import epipolarconsistency_amd as E
    g = np.load(os.path.join(G, "variants_128.npz"))
    for name, (f, post) in _RADON_VARIANTS.items():
        got = E.RadonIntermediate.compute(gpu_ctx, g["image"], 96, 80, filter=f, post_process=post).readback()
        if name == "ramp":
            want = g["radon_ramp_samples"]
            assert np.abs(got.reshape(-1)[g["bins"]] - want).max() <= 1e-6 * np.abs(want).max()
        else:
            assert np.array_equal(_checksum(got), g["radon_%s_checksum" % name]), name
            assert np.array_equal(got.reshape(-1)[g["bins"]], g["radon_%s_samples" % name]), name
    s = small_scan
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, 128, 128) for d in s["dtrs"]]
    m = E.MetricRadonIntermediate(gpu_ctx, s["Ps"], dtrs)
    out = np.zeros(len(g["idx"]), np.float32)
    mean = m.evaluate(g["idx"], out)
    np.testing.assert_allclose(out, g["idx_pairs"], rtol=2e-4)
    assert abs(mean - float(g["idx_mean"])) <= 5e-5 * float(g["idx_mean"])
    assert abs(m.evaluate(set(int(v) for v in g["subset"])) - float(g["subset_mean"])) <= 5e-5 * float(g["subset_mean"])
    # library default (ECC_SAMPLING_AUTO -> reference arithmetic for these few pairs): 1e-6 on every value
    m.setSampling("auto")
    mean = m.evaluate(g["idx"], out)
    np.testing.assert_allclose(out, g["idx_pairs"], rtol=1e-6)
    assert abs(mean - float(g["idx_mean"])) <= 1e-6 * float(g["idx_mean"])
    assert abs(m.evaluate(set(int(v) for v in g["subset"])) - float(g["subset_mean"])) <= 1e-6 * float(g["subset_mean"])
    m.setSampling("polynomial")
    m.setObjectRadius(25.0)
    m.setEpipolarPlaneStep(0.004)
    total, vals = m.evaluate_range(0, 28, want_pairs=True)
    np.testing.assert_allclose(vals, g["param_pairs"], rtol=2e-4)
    assert abs(total / 28 - float(g["param_mean"])) <= 1e-5 * float(g["param_mean"])
