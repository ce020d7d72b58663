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
