"""The reference's launcher seam over the C ABI (Option B): ecc_radon_compute_linear -- what
computeDerivLineIntegrals(tex, n_x, n_y, n_alpha, n_t, filter, post, out_d) does (ref: RadonIntermediate.cpp:12,
RadonIntermediate.cu:149-170), result in the reference's n_t x n_alpha layout --, ecc_dtr_from_device_linear (the reference's
texture upload, ref: RadonIntermediate.cpp:188-196) and ecc_metric_evaluate_external -- what epipolarConsistency(...) does
with the caller's device buffers (ref: EpipolarConsistencyRadonIntermediate.cpp:16-37, .cu:300-409).  All against the class
path and the oracle.  (The C++ form with the reference's argument lists: tests/cpp/test_option_b.cpp.)"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ptr(t):
    return C.c_void_p(t.data_ptr())


@pytest.mark.parametrize("filt,post", [(0, 0), (0, 1), (0, 2), (2, 0), (1, 0)])
def test_radon_compute_linear_is_the_readback_layout(gpu_ctx, oracle_mod, filt, post):
    import torch
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(4)
    n_u, n_v, n_alpha, n_t = 150, 110, 70, 90
    img = rng.uniform(0, 40, (n_v, n_u)).astype(np.float32)
    dev = torch.device("cuda", gpu_ctx.device)
    img_d = torch.from_numpy(img).to(dev)
    out_d = torch.full((n_t * n_alpha + 64,), -7.0, dtype=torch.float32, device=dev)  # exactly n_t * n_alpha floats are written
    torch.cuda.synchronize()
    _lib.check(L.ecc_radon_compute_linear(gpu_ctx._h, _ptr(img_d), n_u, n_v, n_alpha, n_t, filt, post, _ptr(out_d)))
    gpu_ctx.synchronize()
    got = out_d.cpu().numpy()
    assert np.all(got[n_t * n_alpha:] == -7.0)
    got = got[:n_t * n_alpha].reshape(n_t, n_alpha)
    want = E.RadonIntermediate.compute(gpu_ctx, img, n_alpha, n_t, filter=filt, post_process=post).readback()
    assert np.array_equal(got, want)
    ref = oracle_mod.radon(img, n_alpha, n_t, filter=filt, post=post)
    if filt == 1:
        assert np.abs(got - ref).max() <= 1e-6 * np.abs(ref).max()
    else:
        assert np.array_equal(got, ref)
    # and back in: a Radon intermediate made from that device buffer reads back the same bits
    h = C.c_void_p()
    _lib.check(L.ecc_dtr_from_device_linear(gpu_ctx._h, _ptr(out_d), n_alpha, n_t, n_u, n_v, filt, C.byref(h)))
    d = E.RadonIntermediate(gpu_ctx, h)
    assert np.array_equal(d.readback(), got)
    d.close()


@pytest.mark.parametrize("mode", ["auto", "polynomial"])
def test_evaluate_external_reproduces_the_class_path(gpu_ctx, oracle_mod, small_scan, mode):
    import torch
    import epipolarconsistency_amd as E
    from epipolarconsistency_amd import _lib
    L = _lib.lib()
    s = small_scan
    n = len(s["Ps"])
    dev = torch.device("cuda", gpu_ctx.device)
    dtrs = [E.RadonIntermediate.from_host(gpu_ctx, d, s["n_u"], s["n_v"]) for d in s["dtrs"]]
    m = E.MetricRadonIntermediate(gpu_ctx, s["Ps"], dtrs).setSampling(mode)
    want_cost = np.full((n, n), 5.0, np.float32)
    want_mean = m.evaluate(want_cost)
    # the caller's side of the seam: its own host class made the per-view geometry (ref: ...RadonIntermediate.cpp:134-163)
    PinvTs = np.stack([E.host_pinvT(P) for P in s["Ps"]]).astype(np.float32)
    Cs = np.stack([E.host_source_position(P) for P in s["Ps"]]).astype(np.float32)
    Cs_d, PinvTs_d = torch.from_numpy(Cs).to(dev), torch.from_numpy(PinvTs).to(dev)
    out_d = torch.full((n * n,), 5.0, dtype=torch.float32, device=dev)
    K01_d = torch.zeros((28 * 16,), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    radius = m.getObjectRadius()
    _lib.check(L.ecc_metric_evaluate_external(m._h, n, _ptr(Cs_d), _ptr(PinvTs_d), 28, None, _ptr(K01_d), _ptr(out_d),
                                              C.c_float(radius), C.c_float(0.0), 0))
    cost = out_d.cpu().numpy().reshape(n, n)
    assert np.array_equal(cost, want_cost)
    iu = np.triu_indices(n, 1)
    assert abs(cost[iu[1], iu[0]].astype(np.float64).sum() / 28 - want_mean) <= 1e-12 * want_mean  # the host epilogue's mean
    assert np.array_equal(K01_d.cpu().numpy().reshape(28, 16), m.debug_K01(0, 28))
    # index-list form: out_d receives the values; tuples may mix views and Radon intermediates
    idx = np.array([[0, 1, 0, 1], [2, 5, 2, 5], [7, 3, 7, 3], [1, 6, 4, 2]], np.int32)
    idx_d = torch.from_numpy(idx).to(dev)
    vals_d = torch.zeros(4, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    _lib.check(L.ecc_metric_evaluate_external(m._h, n, _ptr(Cs_d), _ptr(PinvTs_d), 4, _ptr(idx_d), None, _ptr(vals_d),
                                              C.c_float(radius), C.c_float(0.0), 0))
    want_vals = np.empty(4, np.float32)
    m.evaluate(idx, want_vals)
    assert np.array_equal(vals_d.cpu().numpy(), want_vals)
    # the call leaves the metric's own parameters and state alone
    assert m.evaluate() == want_mean
    # user-chosen radius / dkappa of the call, against the oracle
    _lib.check(L.ecc_metric_evaluate_external(m._h, n, _ptr(Cs_d), _ptr(PinvTs_d), 28, None, None, _ptr(out_d),
                                              C.c_float(25.0), C.c_float(0.004), 0))
    cost = out_d.cpu().numpy().reshape(n, n)
    ref = oracle_mod.evaluate_all(s["Ps"], s["dtrs"], s["n_u"], s["n_v"], object_radius_mm=25.0, dkappa=0.004)
    np.testing.assert_allclose(cost[iu[1], iu[0]], ref["pairs"], rtol=2e-4 if mode == "polynomial" else 1e-6)
    m.close()
