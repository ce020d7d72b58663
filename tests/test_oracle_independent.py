"""A second, independent statement of the two kernel bodies that the reference holds only as CUDA (R1 radonDerivative,
E3 kernelEpipolarCosistency): written in numpy float32 scalars from the SPECIFICATION in SURVEY.md 8(a)/(c) (rows R1, E3 and
the normative sampling rule), not from oracle/ecc_oracle.c, and compared with the C oracle.  A transcription slip in either
restatement -- an offset, a sign, an axis, the clip order -- shows up here at the 1e-2 level; the R1 statement agrees bit
for bit, the E3 statement to the float rounding of the stored pair value (measured 5e-8).  The pair geometry K01 is taken from the oracle: it is pinned to the reference's own headers elsewhere
(tests/test_oracle_pins.py)."""
import numpy as np

f32 = np.float32
PI_F = f32(3.14159265359)  # the reference's float constant Pi (RadonIntermediate.cu:8, EpipolarConsistencyCommon.hxx)


def _sinf(x):
    return f32(np.sin(np.float64(x)))


def _cosf(x):
    return f32(np.cos(np.float64(x)))


def tex(img, x, y):
    """un-normalised bilinear clamp rule in fp32 with exact fractional weights."""
    H, W = img.shape
    xb, yb = f32(x - f32(0.5)), f32(y - f32(0.5))
    i, j = np.floor(xb), np.floor(yb)
    fx, fy = f32(xb - f32(i)), f32(yb - f32(j))
    i0, i1 = int(min(max(i, 0), W - 1)), int(min(max(i + 1, 0), W - 1))
    j0, j1 = int(min(max(j, 0), H - 1)), int(min(max(j + 1, 0), H - 1))
    one = f32(1)
    r0 = f32(f32((one - fx) * img[j0, i0]) + f32(fx * img[j0, i1]))
    r1 = f32(f32((one - fx) * img[j1, i0]) + f32(fx * img[j1, i1]))
    return f32(f32((one - fy) * r0) + f32(fy * r1))


def radon_bin(img, n_alpha, n_t, ix, iy, derivative=True, post=0):
    """SURVEY.md 8(a) row R1, fp32 in source order.  derivative=False: the plain line integral (Filter::None and the input
    of Filter::Ramp); post 1 / 2: sign-preserving square root / log(1 + .) of the derivative (RadonIntermediate.cu:125-138)."""
    H, W = img.shape
    n_u, n_v = f32(W), f32(H)
    D = f32(np.sqrt(f32(f32(n_u * n_u) + f32(n_v * n_v))))
    alpha = f32(f32(f32(ix) / f32(n_alpha)) - f32(0.5)) * PI_F
    tau = f32(f32(f32(iy) / f32(n_t)) - f32(0.5)) * D
    l0, l1 = f32(-_sinf(alpha)), _cosf(alpha)
    l2 = f32(-tau)
    l2 = f32(l2 + f32(f32(f32(-0.5) * n_u) * l0 - f32(f32(0.5) * n_v) * l1))
    o0, o1 = f32(-l2 * l0), f32(-l2 * l1)
    d0, d1 = l1, f32(-l0)
    with np.errstate(divide="ignore", invalid="ignore"):
        ts = [f32(f32(1 - o0) / d0), f32(f32(n_u - 1 - o0) / d0), f32(f32(1 - o1) / d1), f32(f32(n_v - 1 - o1) / d1)]
    if f32(d0 * d0) < f32(1e-12):
        ts[0], ts[1] = f32(-1e10), f32(1e10)
    if f32(d1 * d1) < f32(1e-12):
        ts[2], ts[3] = f32(-1e10), f32(1e10)
    ts = sorted(ts)
    t, t1 = ts[1], ts[2]
    u, v = f32(o0 + f32(t * d0)), f32(o1 + f32(t * d1))
    if not (u <= n_u and v <= n_v and u >= 0 and v >= 0) or t1 <= t:
        return f32(0)
    o0, o1 = f32(o0 + f32(0.5)), f32(o1 + f32(0.5))       # texel centres
    s, so, step = f32(0), f32(0), f32(0.66)
    if not derivative:
        while t <= t1:
            s = f32(s + tex(img, f32(o0 + f32(t * d0)), f32(o1 + f32(t * d1))))
            t = f32(t + step)
        return f32(s * step)
    o0, o1 = f32(o0 - f32(f32(0.5) * d1)), f32(o1 + f32(f32(0.5) * d0))  # + half a pixel along the normal
    while t <= t1:
        x, y = f32(o0 + f32(t * d0)), f32(o1 + f32(t * d1))
        s = f32(s + tex(img, x, y))
        so = f32(so + tex(img, f32(x + d1), f32(y - d0)))  # the partner line, one pixel along -normal
        t = f32(t + step)
    r = f32(f32(s - so) * step)
    if post == 1:
        return f32(-np.sqrt(f32(-r))) if r < 0 else f32(np.sqrt(r))
    if post == 2:
        return f32(-f32(np.log(np.float64(f32(-r + f32(1)))))) if r < 0 else f32(np.log(np.float64(f32(r + f32(1)))))
    return r


def test_radon_bins_second_statement(oracle_mod):
    rng = np.random.default_rng(3)
    for (H, W), (n_alpha, n_t) in (((40, 56), (36, 30)), ((33, 33), (16, 41))):
        img = rng.uniform(0, 3, size=(H, W)).astype(np.float32)
        want = oracle_mod.radon(img, n_alpha, n_t)
        bins = [(int(a), int(t)) for a, t in zip(rng.integers(0, n_alpha, 60), rng.integers(0, n_t, 60))]
        bins += [(0, 0), (n_alpha // 2, n_t // 2), (n_alpha - 1, n_t - 1), (n_alpha // 2, 0), (0, n_t // 2)]
        nonzero = 0
        for ix, iy in bins:
            got = radon_bin(img, n_alpha, n_t, ix, iy)
            assert got == want[iy, ix], (ix, iy, got, want[iy, ix])
            nonzero += got != 0
        assert nonzero > 30


def test_radon_variants_second_statement(oracle_mod):
    """Filter::None (plain integrals) and the two post-processes of the derivative."""
    rng = np.random.default_rng(8)
    img = rng.uniform(0, 3, size=(37, 45)).astype(np.float32)
    n_alpha, n_t = 24, 28
    bins = [(int(a), int(t)) for a, t in zip(rng.integers(0, n_alpha, 40), rng.integers(0, n_t, 40))]
    plain = oracle_mod.radon(img, n_alpha, n_t, filter=2)
    root = oracle_mod.radon(img, n_alpha, n_t, filter=0, post=1)
    logd = oracle_mod.radon(img, n_alpha, n_t, filter=0, post=2)
    for ix, iy in bins:
        assert radon_bin(img, n_alpha, n_t, ix, iy, derivative=False) == plain[iy, ix]
        assert radon_bin(img, n_alpha, n_t, ix, iy, post=1) == root[iy, ix]
        assert radon_bin(img, n_alpha, n_t, ix, iy, post=2) == logd[iy, ix]
    assert np.count_nonzero(plain) > 100 and (root < 0).any() and (logd < 0).any()


def pair_value(K0, K1, dtr0, dtr1, n_u, n_v):
    """SURVEY.md 8(a) row E3 for one pair (derivative dtrs), fp32 per sample, float64 sum (the oracle's convention)."""
    n_t, n_alpha = dtr0.shape
    D = f32(np.sqrt(np.float64(n_u) * n_u + np.float64(n_v) * n_v))
    range_t = f32(f32(n_t) * f32(np.float64(D) / n_t))
    dk, kmax = f32(K1[6]), f32(K1[7])

    def sample(K, dtr, c, s):
        l = [f32(f32(K[q] * c) + f32(K[3 + q] * s)) for q in range(3)]
        a = f32(f32(np.arctan2(np.float64(l[1]), np.float64(l[0]))) / PI_F)
        if a < 0:
            a = f32(a + f32(2))
        ln = f32(np.sqrt(f32(f32(l[0] * l[0]) + f32(l[1] * l[1]))))
        d = f32(f32(f32(-f32(l[2] / ln)) / range_t) + f32(0.5))
        sign = f32(1)
        if a > 1:
            a, d, sign = f32(a - f32(1)), f32(f32(1) - d), f32(-1)
        return f32(sign * tex(dtr, f32(a * f32(n_alpha)), f32(d * f32(n_t))))

    acc, k = 0.0, 0
    while True:
        kappa = f32(f32(dk * f32(0.5)) + f32(dk * f32(k)))
        if kappa >= kmax:
            break
        c, s = _cosf(kappa), _sinf(kappa)
        vp = f32(sample(K0, dtr0, c, s) - sample(K1, dtr1, c, s))
        vm = f32(sample(K0, dtr0, f32(-c), s) - sample(K1, dtr1, f32(-c), s))
        acc += float(f32(f32(f32(f32(vp * vp) + f32(vm * vm)) * f32(K0[6])) * dk))
        k += 1
    return acc, k


def test_pair_loop_second_statement(oracle_mod, small_scan):
    s = small_scan
    ref = oracle_mod.evaluate_all(s["Ps"], s["dtrs"], s["n_u"], s["n_v"], want_K01=True)
    n = len(s["Ps"])
    q, checked, total_k = 0, 0, 0
    for i in range(n):
        for j in range(i + 1, n):
            if q % 3 == 0:  # every third pair: a few seconds of Python
                K = ref["K01s"][q]
                got, n_k = pair_value(K[:8], K[8:], s["dtrs"][i], s["dtrs"][j], s["n_u"], s["n_v"])
                want = float(ref["pairs"][q])
                assert abs(got - want) <= 2e-7 * abs(want), (i, j, got, want)  # float rounding of the stored pair value
                checked += 1
                total_k += n_k
            q += 1
    assert checked >= 9 and total_k > 500
