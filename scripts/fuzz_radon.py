#!/usr/bin/env python3
"""Randomised bit-exactness sweep of the Radon kernel (GPU box): python scripts/fuzz_radon.py [cases] [seed] [exact|fma|both]
(third argument: the arithmetic mode, ecc_radon_set_arithmetic; "both" alternates per case; each mode against ITS oracle variant)
Random image sizes (odd, tiny, wide, tall), bin grids, filters (derivative / none) and post-processes, image contents
(smooth, noise, constant, sparse) -> np.array_equal against the oracle.  Filter::Ramp goes through a float64
convolution and is compared at 1e-6 of the maximum."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import epipolarconsistency_amd as E  # noqa: E402
import oracle  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
arith = sys.argv[3] if len(sys.argv) > 3 else "exact"
rng = np.random.default_rng(seed)
ctx = E.Context(0)
bad = 0
t0 = time.time()
for c in range(cases):
    n_u = int(rng.choice([3, 17, 40, 64, 97, 128, 200, 255, 320]))
    n_v = int(rng.choice([3, 16, 33, 64, 100, 128, 190, 256]))
    n_alpha = int(rng.choice([5, 16, 33, 64, 96, 130]))
    n_t = int(rng.choice([4, 16, 31, 64, 96, 150]))
    filt = int(rng.choice([E.FILTER_DERIVATIVE, E.FILTER_DERIVATIVE, E.FILTER_NONE, E.FILTER_RAMP]))
    post = int(rng.choice([E.POST_IDENTITY, E.POST_IDENTITY, E.POST_SQUARE_ROOT, E.POST_LOGARITHM]))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        yy, xx = np.mgrid[0:n_v, 0:n_u]
        img = (np.sin(xx * 0.11) * np.cos(yy * 0.07) * 50 + 60).astype(np.float32)
    elif kind == 1:
        img = rng.standard_normal((n_v, n_u)).astype(np.float32) * 100
    elif kind == 2:
        img = np.full((n_v, n_u), float(rng.uniform(-5, 5)), np.float32)
    else:
        img = np.zeros((n_v, n_u), np.float32)
        img[rng.integers(0, n_v, 5), rng.integers(0, n_u, 5)] = 1000.0
    mode = arith if arith != "both" else ("fma" if c % 2 else "exact")
    ctx.setRadonArithmetic(mode)
    n_img = int(rng.integers(1, 4))
    imgs = np.stack([img * (k + 1) for k in range(n_img)])
    dtrs = E.RadonIntermediate.compute_batch(ctx, imgs, n_alpha, n_t, filter=filt, post_process=post)
    ok = True
    for k, d in enumerate(dtrs):
        got = d.readback()
        want = oracle.radon(imgs[k], n_alpha, n_t, filter=filt, post=post, contract=mode == "fma")
        if filt == E.FILTER_RAMP:
            ok = ok and np.abs(got - want).max() <= 1e-6 * max(np.abs(want).max(), 1e-30)
        else:
            ok = ok and np.array_equal(got, want)
        d.close()
    if not ok:
        bad += 1
    print("case %2d: %3dx%3d x%d -> %3dx%3d filter %d post %d kind %d %s: %s" % (c, n_u, n_v, n_img, n_alpha, n_t, filt, post, kind, mode,
                                                                              "ok" if ok else "MISMATCH"), flush=True)
print("%d of %d cases differ, %.1f s" % (bad, cases, time.time() - t0))
sys.exit(1 if bad else 0)
