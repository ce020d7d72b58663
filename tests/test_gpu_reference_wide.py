"""ECC_SAMPLING_REFERENCE from 1024 threads per pair (pairs_reference_wide_kernel, and the one-launch form of
small_eval_kernel.hip) for launches of at most 512 pairs: every pair value has the bits of the 256-thread kernel
pairs_reference_kernel<.., 4>, which launches of 513 ... 2048 pairs keep (the terms are staged in LDS and added in that
kernel's order).  The CPU-path arithmetic itself is pinned against the oracle in test_gpu_metric.py / test_gpu_small_eval.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _metric(gpu_ctx, n, S, B, corr, small, filt=None, seed=11):
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import synthetic
    rng = np.random.default_rng(seed)
    Ps = synthetic.short_scan(n, S, S, 0.308 * 1024 / S)
    kw = {} if filt is None else {"filter": filt}
    base = [E.RadonIntermediate.from_host(gpu_ctx, rng.standard_normal((B, B)).astype(np.float32), S, S, **kw) for _ in range(5)]
    m = E.MetricRadonIntermediate(gpu_ctx, Ps, [base[v % 5] for v in range(n)]).setSampling("reference").setSmallEval(small)
    if corr:
        m.useCorrelation(True)
    return m, base


@pytest.mark.parametrize("corr", [False, True])
@pytest.mark.parametrize("small", [True, False])
def test_wide_launches_have_the_bits_of_the_256_thread_kernel(gpu_ctx, corr, small):
    n = 64  # 2016 pairs: the full range keeps four waves per pair, launches above 512 pairs the 256-thread kernel
    m, base = _metric(gpu_ctx, n, 160, 96, corr, small)
    n_pairs = n * (n - 1) // 2
    _, ref = m.evaluate_range(0, n_pairs, want_pairs=True)  # one launch of 2016 pairs: pairs_reference_kernel<.., 4>
    assert np.isfinite(ref).all() and np.unique(ref).size > 1000
    for first, count in ((0, 1), (5, 100), (100, 192), (300, 193), (700, 512), (1504, 512), (2015, 1)):
        s, v = m.evaluate_range(first, count, want_pairs=True)
        assert np.array_equal(v, ref[first:first + count]), (first, count)
    # index lists (pairs in any order, repeated): the values of the same pairs
    rng = np.random.default_rng(5)
    all_ij = [(i, j) for i in range(n) for j in range(i + 1, n)]
    cost = np.zeros((n, n), np.float32)
    m.evaluate(cost)
    for count in (1, 37, 192, 400, 512):
        pick = [all_ij[k] for k in rng.integers(0, n_pairs, size=count)]
        idx4 = np.array([(i, j, i, j) for i, j in pick], np.int32)
        vals = np.empty(count, np.float32)
        m.evaluate(idx4, vals)
        # cost image of the all-pairs evaluation: cost[j, i] is pair (i, j)'s value (2016 pairs: the 256-thread kernel)
        want = np.array([cost[j, i] for i, j in pick], np.float32)
        assert np.array_equal(vals, want), count
    m.close()
    for d in base:
        d.close()


def test_wide_launch_on_long_pairs_and_plain_line_integrals(gpu_ctx):
    """1448 samples per pair (1024^2 detector): three terms per thread; non-derivative intermediates."""
    import epipolarconsistency_amd as E
    n = 40  # 780 pairs
    m, base = _metric(gpu_ctx, n, 1024, 128, False, True, filt=E.FILTER_NONE)
    n_pairs = n * (n - 1) // 2
    _, ref = m.evaluate_range(0, n_pairs, want_pairs=True)  # 780 pairs: the 256-thread kernel
    for first, count in ((0, 7), (3, 150), (200, 500)):
        _, v = m.evaluate_range(first, count, want_pairs=True)
        assert np.array_equal(v, ref[first:first + count]), (first, count)
    m.close()
    for d in base:
        d.close()
