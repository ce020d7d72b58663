#!/usr/bin/env python3
"""Randomised sweep of MetricDirect (GPU box): python scripts/fuzz_direct.py [cases] [seed]
Random small scans (sizes, perturbed views, object radius, user dkappa, derivative or fan-beam (FBCC) form) -> all-pairs
sum against the oracle: 1e-5 relative for both forms (round 3, 200 cases: 1.4e-15 / 1.1e-15 at most)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import epipolarconsistency_amd as E  # noqa: E402
from epipolarconsistency_amd import geometry, synthetic  # noqa: E402
import oracle  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(seed)
ctx = E.Context(0)
bad = 0
t0 = time.time()
for c in range(cases):
    n = int(rng.integers(2, 6))
    n_u = int(rng.choice([48, 64, 96, 128]))
    n_v = int(rng.choice([48, 64, 80, 128]))
    pixel = 0.308 * 1024.0 / max(n_u, n_v)
    Ps = synthetic.short_scan(n, n_u, n_v, pixel, span_deg=float(rng.choice([200.0, 90.0, 40.0])))
    if rng.integers(0, 2):
        Ps = [P @ geometry.rigid_transform(*(rng.normal(0, 2.0, 3)), *(rng.normal(0, 0.02, 3))) for P in Ps]
    imgs = synthetic.projections_numpy(Ps, n_u, n_v, synthetic.sphere_phantom(seed=int(rng.integers(1, 99)), extent_mm=30,
                                                                             rmin=6, rmax=22))
    imgs = np.ascontiguousarray(imgs + rng.uniform(0, 0.05, imgs.shape).astype(np.float32), np.float32)
    fbcc = bool(rng.integers(0, 2))
    radius = float(rng.choice([0.0, 0.0, 30.0, 70.0]))
    dkappa = float(rng.choice([0.0, 0.0, 0.003, 0.01]))
    m = E.MetricDirect(ctx, Ps, imgs)
    m.setObjectRadius(radius)
    m.setEpipolarPlaneStep(dkappa)
    m.setFanBeamConsistency(fbcc)
    got = m.evaluate()
    want = oracle.direct_evaluate(Ps, imgs, dkappa=dkappa, object_radius_mm=radius, fbcc=fbcc)["sum"]
    rel = abs(got - want) / max(abs(want), 1e-30)
    # (an FBCC weight can divide by zero for a small user radius: the reference's formula, both sides return inf)
    ok = (not np.isfinite(want) and (got == want or (np.isnan(got) and np.isnan(want)))) or rel <= 1e-5
    bad += 0 if ok else 1
    print("case %2d: n=%d %3dx%3d fbcc %d r=%4.1f dk=%.3f: hip %.7g oracle %.7g rel %.2e %s" % (c, n, n_u, n_v, fbcc, radius, dkappa,
                                                                                                   got, want, rel, "ok" if ok else "MISMATCH"), flush=True)
    m.close()
print("%d of %d cases differ, %.1f s" % (bad, cases, time.time() - t0))
sys.exit(1 if bad else 0)
