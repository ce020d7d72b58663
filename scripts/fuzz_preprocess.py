#!/usr/bin/env python3
"""Randomised sweep of the pre-processing kernel (GPU box): python scripts/fuzz_preprocess.py [cases] [seed]
Random image sizes, intensity / low-pass / flip / border / blank settings -> bit-exact against the oracle's
restatement of PreProccess::process; with cosine weighting (projection matrices given) within 2e-7 of the maximum."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import epipolarconsistency_amd as E  # noqa: E402
from epipolarconsistency_amd import synthetic  # noqa: E402
import oracle  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(seed)
ctx = E.Context(0)
bad = 0
t0 = time.time()
for c in range(cases):
    n_u = int(rng.choice([8, 33, 64, 70, 97, 128, 131, 200, 257]))
    n_v = int(rng.choice([8, 31, 64, 65, 100, 128, 190]))
    n = int(rng.integers(1, 4))
    kw = dict(
        normalize=bool(rng.integers(0, 2)), bias=float(rng.choice([0.0, 0.01, -0.2])), scale=float(rng.choice([1.0, 0.5, 3.0])),
        apply_log=bool(rng.integers(0, 2)), gaussian_sigma=float(rng.choice([0.0, 0.5, 1.84, 4.0])),
        half_kernel_width=int(rng.choice([0, 1, 2, 5, 9, 16])), flip_u=bool(rng.integers(0, 2)), flip_v=bool(rng.integers(0, 2)),
        zero=tuple(int(v) for v in rng.choice([0, 1, 3, 12], 4)), feather=tuple(int(v) for v in rng.choice([0, 4, 16, 40], 4)),
        blanks=[tuple(int(v) for v in (rng.integers(-5, n_u), rng.integers(-5, n_v), rng.integers(1, 40), rng.integers(1, 40)))
                for _ in range(int(rng.integers(0, 3)))])
    imgs = rng.uniform(-0.05, 1.5, size=(n, n_v, n_u)).astype(np.float32)
    imgs[0, rng.integers(0, n_v), rng.integers(0, n_u)] = 0.0
    pp = E.PreProccess()
    for k, v in kw.items():
        for ns in (pp.intensity, pp.lowpass, pp.image_geometry, pp.border):
            if hasattr(ns, k):
                setattr(ns, k, list(v) if isinstance(v, tuple) else v)
    with_P = bool(rng.integers(0, 2))
    Ps = synthetic.short_scan(max(n, 2), n_u, n_v, 0.308 * 1024 / n_u)[:n] if with_P else None
    try:
        got = pp.process(ctx, imgs, Ps)
    except E.EccError as e:
        print("case %2d: %s -> rejected: %s" % (c, kw, e), flush=True)
        continue
    ok = True
    worst = 0.0
    for k in range(n):
        want = oracle.preprocess(imgs[k], Ps[k] if with_P else None, **kw)
        if with_P:
            d = np.abs(got[k] - want).max() / max(np.abs(want).max(), 1e-30)
            worst = max(worst, d)
            ok = ok and d <= 2e-7
        else:
            ok = ok and np.array_equal(got[k], want)
    if not ok:
        bad += 1
    print("case %2d: %3dx%3d x%d cos %d %s: %s (%.1e)" % (c, n_u, n_v, n, with_P, kw, "ok" if ok else "MISMATCH", worst), flush=True)
print("%d of %d cases differ, %.1f s" % (bad, cases, time.time() - t0))
sys.exit(1 if bad else 0)
