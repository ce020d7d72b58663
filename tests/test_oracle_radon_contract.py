"""The contracted-arithmetic variant of the Radon intermediate in the oracle (eccor_set_radon_contract(1)) -- the definition
ecc_radon_set_arithmetic(ECC_RADON_FMA) is held to bit for bit on the GPU (tests/test_gpu_radon_fma.py).

ref: RadonIntermediate.cu:118-123 (o + t * d: nvcc contracts it), LibUtilsCuda/CudaBindlessTexture.cpp:25-39 (the GPU build
interpolates in texture hardware: the unfused (1 - f) * a + f * b is the CPU reading of the source, nothing the
reference's GPU executes).  The contracted variant is tied to the normative one at the level of the METRIC here."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _checksum(a):
    a = np.ascontiguousarray(a, np.float32)
    return np.array([a.astype(np.float64).sum(), np.abs(a).astype(np.float64).sum(),
                     float(np.bitwise_xor.reduce(a.view(np.uint32).reshape(-1)))])


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "radon_contract.npz"))


def test_contract_flag_is_scoped_and_restored(oracle_mod, small_scan):
    L = oracle_mod.lib()
    L.eccor_get_radon_contract.restype = int
    im = small_scan["imgs"][2]
    a = oracle_mod.radon(im, 48, 40)
    b = oracle_mod.radon(im, 48, 40, contract=True)
    assert L.eccor_get_radon_contract() == 0  # the switch does not leak into later calls
    assert np.array_equal(a, oracle_mod.radon(im, 48, 40))
    assert not np.array_equal(a, b)
    bins = np.arange(0, 48 * 40, 7, dtype=np.int32)
    assert np.array_equal(oracle_mod.radon_bins(im, 48, 40, bins, contract=True), b.reshape(-1)[bins])
    assert np.array_equal(oracle_mod.radon_bins(im, 48, 40, bins), a.reshape(-1)[bins])


def test_contracted_variant_against_its_golden(oracle_mod, small_scan, golden):
    dtrs = [oracle_mod.radon(im, 96, 96, contract=True) for im in small_scan["imgs"]]
    assert np.array_equal(np.stack([_checksum(d) for d in dtrs]), golden["synthetic8_dtr_checksums"])
    res = oracle_mod.evaluate_all(small_scan["Ps"], dtrs, 128, 128)
    assert res["mean"] == float(golden["synthetic8_mean"]) and np.array_equal(res["pairs"], golden["synthetic8_pairs"])
    v = np.load(os.path.join(HERE, "golden", "variants_128.npz"))
    for name, (f, post) in dict(deriv=(0, 0), deriv_sqrt=(0, 1), deriv_log=(0, 2), plain=(2, 0), ramp=(1, 0)).items():
        d = oracle_mod.radon(v["image"], 96, 80, filter=f, post=post, contract=True)
        assert np.array_equal(_checksum(d), golden["variants_%s_checksum" % name]), name
        assert np.array_equal(d.reshape(-1)[v["bins"]], golden["variants_%s_samples" % name]), name


def test_contracted_and_exact_variant_agree_on_the_metric(oracle_mod, small_scan, golden):
    """Metric-level tie, CPU-sized cases (the BASELINE configs are tied on the GPU, where both modes are bit-identical to
    their oracle variant: tests/test_gpu_radon_fma.py).  Measured: 8-view scan 6.8e-7; the example pair at 256 x 190 --
    ONE pair, nothing averages -- 2.3e-6, at its native size 7.4e-6 (the fixture's numbers, made by make_golden.py).
    A single pair's value moves by 2e-5 (median) under ANY change of fp32 rounding order, so the bar for
    means over pairs is 2e-6 and for the single example pair north_star's 1e-5."""
    exact = small_scan["dtrs"]
    contr = [oracle_mod.radon(im, 96, 96, contract=True) for im in small_scan["imgs"]]
    scale = max(np.abs(d).max() for d in exact)
    dev = max(np.abs(a - b).max() for a, b in zip(exact, contr)) / scale
    assert 0 < dev < 2e-5  # a different rounding of the same sums, not a different function
    m0 = oracle_mod.evaluate_all(small_scan["Ps"], exact, 128, 128)["mean"]
    m1 = oracle_mod.evaluate_all(small_scan["Ps"], contr, 128, 128)["mean"]
    assert abs(m1 - m0) / abs(m0) < 2e-6
    for k in ("pair256", "native"):
        rel = abs(float(golden[k + "_mean"]) - float(golden[k + "_mean_exact"])) / abs(float(golden[k + "_mean_exact"]))
        assert rel < 1e-5, (k, rel)
        assert float(golden[k + "_max_bin_dev_rel"]) < 5e-5


def test_example_pair_256_contracted_golden(oracle_mod, golden):
    g = np.load(os.path.join(HERE, "golden", "example_pair_256.npz"))
    dtrs = [oracle_mod.radon(im, int(g["n_alpha"]), int(g["n_t"]), contract=True) for im in g["images"]]
    assert np.array_equal(np.stack([_checksum(d) for d in dtrs]), golden["pair256_dtr_checksums"])
    assert np.array_equal(np.stack([d.reshape(-1)[g["sample_bins"]] for d in dtrs]), golden["pair256_dtr_samples"])
    assert oracle_mod.evaluate_all(list(g["Ps"]), dtrs, 256, 190)["mean"] == float(golden["pair256_mean"])
