"""Pins the CPU oracle (oracle/ecc_oracle.c) before anything trusts it:
  * against the reference's own host-compilable headers (oracle/_ref/libecc_ref.so, built by
    oracle/Makefile from /root/reference/code/LibEpipolarConsistency/EpipolarConsistencyCommon.hxx and
    LibUtilsCuda/culaut/*.hxx) -- E1/E2 geometry, bit for bit in float32;
  * against the known-answer scalars of the example pair recorded in SURVEY.md 8c;
  * against residual KATs in the style of the reference's disabled TestCudaUtils.cpp:39-57.
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
