"""Pins the CPU oracle (oracle/ecc_oracle.c) before anything trusts it:
  * against the reference's own host-compilable headers (oracle/_ref/libecc_ref.so, built by
    oracle/Makefile from /root/reference/code/LibEpipolarConsistency/EpipolarConsistencyCommon.hxx and
    LibUtilsCuda/culaut/*.hxx) -- E1/E2 geometry, bit for bit in float32;
  * against the known-answer scalars of the example pair recorded in SURVEY.md 8c;
  * against residual KATs in the style of the reference's disabled TestCudaUtils.cpp:39-57.
  * the pre-processing low-pass (Gaussian taps and the separable convolution with its dropped last tap) and the host
    image interpolation against the reference's header-only NRRD library (nrrd_lowpass.hxx, nrrd_image_view.hxx),
    compiled into the same oracle/_ref/libecc_ref.so -- bit for bit;
  * the NORMATIVE variant 0 of the oracle (what every GPU test is held to) against the PINNED variant 2 at the level
    of the metric (64-view short scan): the two differ only by the rounding of four elementary functions.
The kernel bodies R1/E3 have no reference vectors (parity unpinned, see oracle header): they are
covered by property tests (test_oracle_properties.py) and committed goldens (test_golden.py).
"""
import numpy as np
import pytest

P000 = np.array([[-506.148, -0, -3532.97, 376726], [-384, -3532.97, -0, 285811], [-1, -0, -0, 744.3]])
P040 = np.array([[-1975.44, -0, 2972.49, 376726], [286.441, -3532.97, 255.749, 285811],
                 [0.745941, -0, 0.666012, 744.3]])  # ref: config/example_data/proj0{00,40}.nrrd "Projection Matrix"


def _need_ref(oracle_mod):
    r = oracle_mod.ref()
    if r is None:
        pytest.skip("oracle/_ref/libecc_ref.so not built and /root/reference absent")
    return r


@pytest.fixture
def libm_variant(oracle_mod):
    """The reference headers call the platform's float libm; variant 2 makes the oracle do the
    same so the comparison is bit for bit (the normative variant 0 swaps in correctly rounded
    sinf/cosf/atan2f/asinf and nothing else)."""
    oracle_mod.set_variant(2)
    yield
    oracle_mod.set_variant(0)


def _random_Ps(k, seed=0):
    from epipolarconsistency_amd import synthetic, geometry
    rng = np.random.default_rng(seed)
    Ps = synthetic.short_scan(k, 512, 384, 0.5)
    out = []
    for P in Ps:
        T = geometry.rigid_transform(*rng.uniform(-20, 20, 3), *rng.uniform(-0.3, 0.3, 3))
        out.append(P @ T * rng.uniform(0.5, 3.0))
    return out


def test_e1_matches_reference_headers_bitwise(oracle_mod):
    _need_ref(oracle_mod)
    for P in [P000, P040] + _random_Ps(40):
        a, b = oracle_mod.pinvT(P), oracle_mod.pinvT(P, use_ref=True)
        assert np.array_equal(a, b), (a, b)
        c, d = oracle_mod.source_position(P), oracle_mod.source_position(P, use_ref=True)
        assert np.array_equal(c, d), (c, d)


def test_get_ij_matches_reference(oracle_mod):
    _need_ref(oracle_mod)
    for n in (2, 3, 4, 5, 17, 64):
        seen = set()
        for ij in range(n * (n - 1) // 2):
            a = oracle_mod.get_ij(ij, n)
            assert a == oracle_mod.get_ij(ij, n, use_ref=True)
            assert a[0] < a[1] < n
            seen.add(a)
        assert len(seen) == n * (n - 1) // 2
    assert oracle_mod.get_ij(5, 4) == (2, 3)


def test_computeK01_and_line_mapping_match_reference(oracle_mod, libm_variant):
    _need_ref(oracle_mod)
    Ps = [P000, P040] + _random_Ps(12, seed=3)
    rng = np.random.default_rng(1)
    for a in range(0, len(Ps), 2):
        P0, P1 = Ps[a], Ps[a + 1]
        args = (512.0, 380.0, oracle_mod.source_position(P0), oracle_mod.source_position(P1),
                oracle_mod.pinvT(P0), oracle_mod.pinvT(P1), np.float32(106.75), np.float32(2550.4))
        for dk in (0.0, 0.003):
            K0, K1 = oracle_mod.computeK01(*args, dk)
            R0, R1 = oracle_mod.computeK01(*args, dk, use_ref=True)
            assert np.array_equal(K0, R0) and np.array_equal(K1, R1)
        for _ in range(200):
            line = rng.normal(size=3).astype(np.float32) * np.float32([1, 1, 500])
            (la, ma), (lb, mb) = oracle_mod.line_to_sample_dtr(line, 1275.2), \
                oracle_mod.line_to_sample_dtr(line, 1275.2, use_ref=True)
            assert ma == mb and np.array_equal(la, lb)


def test_example_pair_known_answers(oracle_mod):
    """scalars obtained from the reference headers for the example pair."""
    n_u, n_v, n_t = 1024, 760, 768
    r = oracle_mod.object_radius(P000, n_u, n_v)
    assert abs(r - 106.75) < 5e-3
    C0, C1 = oracle_mod.source_position(P000), oracle_mod.source_position(P040)
    np.testing.assert_allclose(C0, [744.3, -5.66e-5, 1.23e-5, 1], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(C1, [-555.205, -5.76e-5, -495.712, 1], rtol=1e-5, atol=1e-7)
    step_t = np.float32(np.sqrt(n_u ** 2 + n_v ** 2) / n_t)
    K0, K1 = oracle_mod.computeK01(n_u * 0.5, n_v * 0.5, C0, C1, oracle_mod.pinvT(P000), oracle_mod.pinvT(P040),
                                   np.float32(r), np.float32(n_t) * step_t * np.float32(2))
    np.testing.assert_allclose(K0[[1, 2, 3, 5, 6, 7]], [-1, 3.99973106, 0.934329152, -1253.72327, 265.277008, 2.41274428],
                               rtol=2e-6)
    np.testing.assert_allclose(K1[[1, 2, 3, 5, 6, 7]], [-1, 3.99973106, -0.934328914, -1264.65845, 3.24763969e-04,
                                                        0.414144218], rtol=2e-6)
    assert abs(K0[0]) < 1e-6 and abs(K0[4]) < 1e-6 and abs(K1[0]) < 1e-6 and abs(K1[4]) < 1e-6
    # N_kappa = #{k : dkappa (k + 1/2) < kappa_max} = 1275
    k = np.arange(4096, dtype=np.float32)
    assert int(np.sum(K1[6] * np.float32(0.5) + K1[6] * k < K1[7])) == 1275


def test_residual_kats(oracle_mod):
    """||P C|| ~ 0 and ||P P^+ - I||_F ~ 0 (form of ref: LibUtilsCuda/TestCudaUtils.cpp:39-57)."""
    for P in [P000, P040] + _random_Ps(20, seed=9):
        C = oracle_mod.source_position(P).astype(np.float64)
        assert np.linalg.norm(P @ C) / (np.linalg.norm(P) * np.linalg.norm(C)) < 1e-6
        PinvT = oracle_mod.pinvT(P).astype(np.float64).reshape(4, 3).T  # 3x4 column-major -> (P^+)^T
        assert np.linalg.norm(P @ PinvT.T - np.eye(3)) < 1e-4


def test_weighting_matches_reference(oracle_mod):
    r = _need_ref(oracle_mod)
    for x in (-2.0, -1.0, -0.5, 0.0, 0.3, 1.0, 1.5):
        xx = np.float32(x) * np.float32(x)
        want = 0.0 if abs(x) > 1 else float(np.float32(1) - 2 * xx + xx * xx)
        assert abs(r.ref_weighting(x) - want) < 1e-6


def test_normative_variant_differs_only_by_elementary_function_rounding(oracle_mod):
    """Variant 0 (correctly rounded functions) vs variant 2 (glibc float functions): same K01 up
    to the last ulp of the two entries that come out of atan2f/asinf."""
    args = (512.0, 380.0, oracle_mod.source_position(P000), oracle_mod.source_position(P040),
            oracle_mod.pinvT(P000), oracle_mod.pinvT(P040), np.float32(106.75), np.float32(2550.4))
    K0, K1 = oracle_mod.computeK01(*args)
    oracle_mod.set_variant(2)
    try:
        L0, L1 = oracle_mod.computeK01(*args)
    finally:
        oracle_mod.set_variant(0)
    assert np.array_equal(K0[:7], L0[:7]) and np.array_equal(K1[:6], L1[:6])
    np.testing.assert_allclose(K0[7], L0[7], rtol=2e-7)
    np.testing.assert_allclose(K1[6:], L1[6:], rtol=2e-7)


def test_gaussian_kernel_matches_reference_bitwise(oracle_mod):
    """eccor_gaussian_kernel vs NRRD::gaussianKernel (ref: HeaderOnly/NRRD/nrrd_lowpass.hxx:19-34)."""
    r = _need_ref(oracle_mod)
    for sigma, k in ((1.84, 5), (1.0, 2), (0.7, 1), (3.3, 9), (2.5, 12), (10.0, 3)):
        want = np.zeros(2 * k + 1, np.float64)
        r.ref_gaussian_kernel(sigma, k, want)
        got = oracle_mod.gaussian_kernel(sigma, k)
        assert np.array_equal(got, want), (sigma, k)
        assert abs(want.sum() - 1.0) < 1e-15


def test_lowpass_stage_matches_reference_bitwise(oracle_mod):
    """The low-pass stage of eccor_preprocess vs NRRD::lowpass2D (ref: nrrd_lowpass.hxx:183-190 -> convolve2D :45-79:
    both passes run o = -k .. k-1 -- the last tap is dropped --, both use kernelx, float64 sums rounded to float per
    pass, clamp addressing).  Every other stage of the pre-processing is switched off; images are non-negative so the
    intensity stage (x * 1 + 0, negatives to zero) is the identity."""
    r = _need_ref(oracle_mod)
    rng = np.random.default_rng(21)
    for (h, w), sigma, k in (((50, 70), 1.84, 5), ((33, 17), 1.0, 2), ((7, 9), 2.5, 6), ((64, 64), 0.8, 3), ((5, 40), 1.84, 5)):
        img = rng.uniform(0, 100, (h, w)).astype(np.float32)
        img[rng.integers(0, h, 3), rng.integers(0, w, 3)] = 1e4  # isolated peaks: every tap position matters
        want = img.copy()
        r.ref_lowpass2D(want, w, h, sigma, k)
        got = oracle_mod.preprocess(img, gaussian_sigma=sigma, half_kernel_width=k, zero=(0,) * 4, feather=(0,) * 4)
        assert np.array_equal(got, want), (h, w, sigma, k)
        assert not np.array_equal(got, img)


def test_tex2d_matches_reference_image_view_where_the_rules_coincide(oracle_mod):
    """eccor_tex2d (texel centres at i + 0.5, float32) vs NRRD::ImageView::operator()(x, y) (texel centres at integers,
    float64; ref: HeaderOnly/NRRD/nrrd_image_view.hxx:189-210) at coordinates where both are exact: integer-valued
    texels, fractions that are multiples of 1/8, positions inside [0, n - 1] (the reference truncates negative
    coordinates towards zero and extrapolates there; the texture rule clamps)."""
    r = _need_ref(oracle_mod)
    rng = np.random.default_rng(4)
    h, w = 13, 17
    img = rng.integers(-500, 500, (h, w)).astype(np.float32)
    n = 0
    for y8 in range(0, 8 * (h - 1) + 1, 3):
        for x8 in range(0, 8 * (w - 1) + 1, 5):
            x, y = x8 / 8.0, y8 / 8.0
            want = r.ref_image_view_at(img, w, h, x, y)
            got = oracle_mod.tex2d(img, x + 0.5, y + 0.5)
            assert float(got) == want, (x, y, got, want)
            n += 1
    assert n > 500
    # on the texel centres themselves both return the texel
    for (i, j) in ((0, 0), (w - 1, h - 1), (3, 7), (w - 1, 0)):
        assert oracle_mod.tex2d(img, i + 0.5, j + 0.5) == img[j, i] == r.ref_image_view_at(img, w, h, float(i), float(j))


def test_normative_and_pinned_variant_agree_on_the_metric(oracle_mod):
    """Config-2 geometry (64-view short scan) with small images: the mean over all 2016 pairs of the
    normative variant 0 (correctly rounded sinf / cosf / atan2f / asinf; what the GPU is compared with) and of variant 2
    (the platform's float libm; the one pinned bit for bit against the reference's headers above) agree to 5e-6
    relative, and single pairs to the fp32 noise floor of the sample positions."""
    from epipolarconsistency_amd import synthetic
    n, S, B = 64, 128, 96
    Ps = synthetic.short_scan(n, S, S, 0.616 * 512 / S)
    imgs = synthetic.projections_numpy(Ps, S, S, synthetic.sphere_phantom())
    dtrs = [oracle_mod.radon(im, B, B) for im in imgs]
    a = oracle_mod.evaluate_all(Ps, dtrs, S, S)
    oracle_mod.set_variant(2)
    try:
        b = oracle_mod.evaluate_all(Ps, dtrs, S, S)
    finally:
        oracle_mod.set_variant(0)
    assert len(a["pairs"]) == 2016
    assert abs(a["mean"] - b["mean"]) <= 5e-6 * abs(a["mean"]), (a["mean"], b["mean"])
    assert a["mean"] != b["mean"] or np.array_equal(a["pairs"], b["pairs"])
    np.testing.assert_allclose(a["pairs"], b["pairs"], rtol=2e-3)
    assert np.median(np.abs(a["pairs"] - b["pairs"]) / np.abs(a["pairs"])) < 5e-5
