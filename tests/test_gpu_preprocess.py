"""Projection pre-processing on the device against the oracle's restatement of
PreProccess::process + apply_weight_cos_principal_ray (ref: Gui/PreProccess.cpp:57-166)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _images(shape, n=3, seed=3):
    rng = np.random.default_rng(seed)
    return rng.uniform(0.05, 1.0, size=(n,) + shape).astype(np.float32)


CASES = [
    dict(),                                                              # the reference's defaults
    dict(apply_log=True, scale=0.9, bias=0.01),
    dict(normalize=True, scale=2.0, apply_log=True),
    dict(flip_u=True), dict(flip_v=True, flip_u=True),
    dict(zero=(0, 3, 2, 0), feather=(5, 0, 7, 30)),
    dict(blanks=[(5, 4, 20, 9), (-3, 30, 8, 500)]),
    dict(gaussian_sigma=0.0), dict(half_kernel_width=1), dict(gaussian_sigma=3.0, half_kernel_width=16),
    dict(gaussian_sigma=0.7, half_kernel_width=2, zero=(0, 0, 0, 0), feather=(0, 0, 0, 0)),
]


def _set(pp, kw):
    for k, v in kw.items():
        for ns in (pp.intensity, pp.lowpass, pp.image_geometry, pp.border):
            if hasattr(ns, k):
                setattr(ns, k, list(v) if isinstance(v, tuple) else v)


@pytest.mark.parametrize("shape", [(40, 56), (97, 131), (64, 64)])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_process_bit_exact(gpu_ctx, oracle_mod, small_scan, shape, case):
    import epipolarconsistency_amd as E
    kw = CASES[case]
    imgs = _images(shape)
    imgs[1, 3, 4] = -2.0  # negative -> 0 (and log of a negative -> NaN -> 0)
    imgs[2, 7, 1] = 0.0   # log(0) -> inf -> 0
    pp = E.PreProccess()
    _set(pp, kw)
    got = pp.process(gpu_ctx, imgs)
    for k in range(len(imgs)):
        want = oracle_mod.preprocess(imgs[k], **kw)
        assert np.array_equal(got[k], want), (kw, np.abs(got[k] - want).max())


def test_cosine_weight_and_in_place_on_device(gpu_ctx, oracle_mod, small_scan):
    import torch
    import epipolarconsistency_amd as E
    s = small_scan
    imgs = np.ascontiguousarray(s["imgs"][:4], np.float32) + 0.25
    Ps = [p.copy() for p in s["Ps"][:4]]
    Ps[2] = np.zeros((3, 4))  # an all-zero matrix skips the weighting of that view (ref: PreProccess.cpp:149)
    pp = E.PreProccess()
    want = np.stack([oracle_mod.preprocess(imgs[k], Ps[k]) for k in range(4)])
    host = pp.process(gpu_ctx, imgs, Ps)
    assert np.abs(host - want).max() <= 2e-7 * np.abs(want).max()
    assert np.array_equal(host[2], oracle_mod.preprocess(imgs[2]))
    # device tensors, in place and out of place
    t = torch.from_numpy(imgs).cuda()
    o = torch.empty_like(t)
    pp.process(gpu_ctx, t, Ps, out=o)
    assert np.array_equal(o.cpu().numpy(), host) and np.array_equal(t.cpu().numpy(), imgs)
    pp.process(gpu_ctx, t, Ps)
    assert np.array_equal(t.cpu().numpy(), host)
    # weighting alone
    w = pp.apply_weight_cos_principal_ray(gpu_ctx, imgs, Ps)
    want_w = np.stack([oracle_mod.preprocess(imgs[k], Ps[k], process=False) for k in range(4)])
    assert np.abs(w - want_w).max() <= 2e-7 * np.abs(want_w).max()
    # ... is not affected by the fields of process(): the flips in particular belong to process() alone
    # (ref: Gui/PreProccess.cpp:123-136 against :146-166)
    pf = E.PreProccess()
    pf.image_geometry.flip_u = pf.image_geometry.flip_v = True
    pf.intensity.scale, pf.border.zero = 3.0, [9, 9, 9, 9]
    assert np.array_equal(pf.apply_weight_cos_principal_ray(gpu_ctx, imgs, Ps), w)
    # intrinsics agree with the oracle's
    a = np.array(oracle_mod.intrinsics(Ps[0]))
    from epipolarconsistency_amd.api import host_intrinsics
    assert np.allclose(host_intrinsics(Ps[0]), a, rtol=2e-7)


def test_preprocess_feeds_radon(gpu_ctx, oracle_mod, small_scan):
    """pre-process -> Radon intermediate, both on the device, equals the oracle's chain bit for bit."""
    import torch
    import epipolarconsistency_amd as E
    s = small_scan
    imgs = np.ascontiguousarray(s["imgs"][:2], np.float32)
    pp = E.PreProccess()
    t = torch.from_numpy(imgs).cuda()
    pp.process(gpu_ctx, t)
    dtrs = E.RadonIntermediate.compute_batch(gpu_ctx, t, 48, 40)
    for k in range(2):
        want = oracle_mod.radon(oracle_mod.preprocess(imgs[k]), 48, 40)
        assert np.array_equal(dtrs[k].readback(), want)


def test_preprocess_errors(gpu_ctx):
    import epipolarconsistency_amd as E
    pp = E.PreProccess()
    pp.lowpass.half_kernel_width = 17
    with pytest.raises(E.EccError) as ei:
        pp.process(gpu_ctx, _images((20, 20)))
    assert ei.value.code == 5
    pp.lowpass.half_kernel_width = 5
    pp.border.zero = [-1, 0, 0, 0]
    with pytest.raises(E.EccError):
        pp.process(gpu_ctx, _images((20, 20)))


def test_preprocess_overlap_rejected_and_arena_reused(gpu_ctx, oracle_mod):
    """Device form: `out` identical to `images` (in place, through the context's scratch stack) or disjoint; a partially
    overlapping output would race with the halo reads and is refused.  Repeated calls of different sizes reuse / grow the
    context's arena and stay bit-exact."""
    import torch
    import epipolarconsistency_amd as E
    pp = E.PreProccess()
    big = torch.zeros((5, 40, 48), dtype=torch.float32, device="cuda")
    with pytest.raises(E.EccError) as ei:
        pp.process(gpu_ctx, big[0:4], out=big[1:5])
    assert "overlaps" in str(ei.value)
    rng = np.random.default_rng(9)
    for shape in ((3, 40, 48), (2, 33, 21), (6, 64, 80), (3, 40, 48)):
        imgs = rng.uniform(0.1, 2.0, size=shape).astype(np.float32)
        want = np.stack([oracle_mod.preprocess(im) for im in imgs])
        t = torch.from_numpy(imgs).cuda()
        pp.process(gpu_ctx, t)  # in place, asynchronous on the context's stream
        gpu_ctx.synchronize()
        assert np.array_equal(t.cpu().numpy(), want)
        assert np.array_equal(pp.process(gpu_ctx, imgs), want)  # host form


def test_normalize_maximum_and_nans(gpu_ctx, oracle_mod):
    """Intensity/Normalize: the maximum is found by several workgroups per image; NaNs never win the search and a NaN
    FIRST pixel stays the 'maximum' (ref: Gui/PreProccess.cpp:68-71) -- bit-identical to the oracle, sizes that do not
    divide into the 32 parts evenly included."""
    import epipolarconsistency_amd as E
    rng = np.random.default_rng(5)
    for shape in ((3, 70, 90), (2, 5, 7), (1, 1, 33)):
        imgs = rng.uniform(0.0, 9.0, size=shape).astype(np.float32)
        imgs[0, shape[1] // 2, shape[2] // 3] = 40.0     # the maximum somewhere inside
        imgs[-1, shape[1] - 1, shape[2] - 1] = 77.0      # ... and in the very last pixel
        if shape[0] > 1:
            imgs[1, 0, 1 % shape[2]] = np.nan             # ignored by the search
        if shape[0] > 2:
            imgs[2, 0, 0] = np.nan                        # first pixel: everything becomes NaN -> 0
        pp = E.PreProccess()
        pp.intensity.normalize = True
        pp.intensity.scale = 3.0
        pp.lowpass.half_kernel_width = 0
        pp.border.zero, pp.border.feather = [0, 0, 0, 0], [0, 0, 0, 0]
        got = pp.process(gpu_ctx, imgs)
        for k in range(shape[0]):
            want = oracle_mod.preprocess(imgs[k], normalize=True, scale=3.0, half_kernel_width=0, zero=(0, 0, 0, 0),
                                         feather=(0, 0, 0, 0))
            assert np.array_equal(got[k], want), (shape, k)
        assert got[0].max() == np.float32(3.0) or shape[0] == 1
        if shape[0] > 2:
            assert not got[2].any()
