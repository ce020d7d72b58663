#!/usr/bin/env python3
"""Randomised parity sweep (GPU box): python scripts/fuzz_parity.py [cases] [seed]
Random scan geometries (sizes, bin grids, perturbed / rotated views, object radius, user dkappa, derivative or plain
dtrs, random dtr contents) -> pair values and mean of the HIP path against the oracle, in two sampling modes:
  * "polynomial" (the throughput path; these problems are far too small to average fp32 position noise out): exit code 1
    if a mean is off by more than 1e-5 x max(1, 30 / sqrt(n_pairs)) relative or a pair by more than 2e-3;
  * "auto" (the library default; <= 512 pairs -> the CPU path's own arithmetic): mean AND every pair within 1e-5.
Prints the worst cases."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import epipolarconsistency_amd as E  # noqa: E402
from epipolarconsistency_amd import geometry, synthetic  # noqa: E402
import oracle  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rng = np.random.default_rng(seed)
ctx = E.Context(0)
worst_mean, worst_pair, worst_auto, bad = 0.0, 0.0, 0.0, 0
t_start = time.time()
for c in range(cases):
    n = int(rng.integers(2, 14))
    n_u = int(rng.choice([64, 96, 128, 200, 256]))
    n_v = int(rng.choice([64, 96, 128, 160, 256]))
    n_alpha = int(rng.choice([48, 64, 96, 128, 192]))
    n_t = int(rng.choice([48, 64, 96, 128, 192]))
    pixel = 0.308 * 1024.0 / max(n_u, n_v) * float(rng.uniform(0.7, 1.3))
    span = float(rng.choice([200.0, 360.0, 90.0, 30.0]))
    Ps = synthetic.short_scan(n, n_u, n_v, pixel, span_deg=span)
    kind = int(rng.integers(0, 4))
    if kind == 1:  # every view perturbed by a rigid motion
        Ps = [P @ geometry.rigid_transform(*(rng.normal(0, 3.0, 3)), *(rng.normal(0, 0.03, 3))) for P in Ps]
    elif kind == 2:  # detector rotated in its plane (epipolar lines far from horizontal)
        a = float(rng.uniform(-1.5, 1.5))
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        Tc = np.array([[1, 0, n_u / 2], [0, 1, n_v / 2], [0, 0, 1.0]])
        Ps = [Tc @ R @ np.linalg.inv(Tc) @ P for P in Ps]
    derivative = bool(rng.integers(0, 4) != 0)
    dtrs_h = [rng.standard_normal((n_t, n_alpha)).astype(np.float32) * 10 + (0 if derivative else 50) for _ in range(n)]
    # smooth them a little: white noise makes the pair values all noise
    dtrs_h = [(d + np.roll(d, 1, 0) + np.roll(d, 1, 1) + np.roll(d, -1, 0)).astype(np.float32) for d in dtrs_h]
    filt = E.FILTER_DERIVATIVE if derivative else E.FILTER_NONE
    dtrs = [E.RadonIntermediate.from_host(ctx, d, n_u, n_v, filter=filt) for d in dtrs_h]
    m = E.MetricRadonIntermediate(ctx, Ps, dtrs).setSampling("polynomial")
    radius = float(rng.choice([0.0, 0.0, 20.0, 80.0, 400.0]))
    dkappa = float(rng.choice([0.0, 0.0, 0.002, 0.01]))
    m.setObjectRadius(radius)
    m.setEpipolarPlaneStep(dkappa)
    n_pairs = n * (n - 1) // 2
    total, vals = m.evaluate_range(0, n_pairs, want_pairs=True)
    want = oracle.evaluate_all(Ps, dtrs_h, n_u, n_v, object_radius_mm=radius, dkappa=dkappa, is_derivative=derivative)
    ok_pairs = np.isfinite(want["pairs"])
    # pairs of views with (numerically) the same source position have no baseline: computeK01 normalises rounding
    # noise (ref: EpipolarConsistencyCommon.hxx:115-129) and every implementation returns its own noise -- e.g. the
    # first and the last view of a 360-degree scan.  They are left out of the comparison.
    Cs = [E.host_source_position(P)[:3].astype(np.float64) for P in Ps]
    for q in range(n_pairs):
        i, j = E.get_ij(q, n)
        if np.linalg.norm(Cs[i] - Cs[j]) < 1e-4 * np.linalg.norm(Cs[i]):
            ok_pairs[q] = False
    if not np.isfinite(want["pairs"]).all():  # 0/0 geometry (baseline through the origin): both sides must agree on which
        assert np.array_equal(np.isfinite(vals), np.isfinite(want["pairs"])), "finite-ness differs"
    ref = want["pairs"][ok_pairs].astype(np.float64)
    got = vals[ok_pairs].astype(np.float64)
    scale = np.maximum(np.abs(ref), 1e-3 * np.abs(ref).max() if ref.size else 1.0)
    rel_pair = float(np.max(np.abs(got - ref) / scale)) if ref.size else 0.0
    rel_mean = abs(got.mean() - ref.mean()) / abs(ref.mean()) if ref.size else 0.0
    degs = [r["degree"] for r in m.debug_polynomials(0, n_pairs)]
    flag = ""
    # the library's default mode on the same problem: strict
    _, vals_a = m.setSampling("auto").evaluate_range(0, n_pairs, want_pairs=True)
    m.setSampling("polynomial")
    got_a = vals_a[ok_pairs].astype(np.float64)
    rel_pair_a = float(np.max(np.abs(got_a - ref) / scale)) if ref.size else 0.0
    rel_mean_a = abs(got_a.mean() - ref.mean()) / abs(ref.mean()) if ref.size else 0.0
    worst_auto = max(globals().get("worst_auto", 0.0), rel_pair_a, rel_mean_a)
    if rel_mean_a > 1e-5 or rel_pair_a > 1e-5:
        bad += 1
        flag += "  <-- DEFAULT MODE OUT OF TOLERANCE (mean %.2e pair %.2e)" % (rel_mean_a, rel_pair_a)
    if rel_mean > 1e-5 * max(1.0, 30.0 / np.sqrt(max(n_pairs, 1))) or rel_pair > 2e-3:
        bad += 1
        flag = "  <-- OUT OF TOLERANCE"
        K01 = m.debug_K01(0, n_pairs)
        for q in np.argsort(-np.abs(got - ref) / scale)[:4]:
            i, j = E.get_ij(int(np.flatnonzero(ok_pairs)[q]), n)
            print("   pair (%d,%d): hip %.7g oracle %.7g  baseline distance %.4g mm, kappa_max %.5f, dkappa %.3g, degree %d"
                  % (i, j, got[q], ref[q], K01[np.flatnonzero(ok_pairs)[q]][6], K01[np.flatnonzero(ok_pairs)[q]][15],
                     K01[np.flatnonzero(ok_pairs)[q]][14], degs[np.flatnonzero(ok_pairs)[q]]))
    # explicit index tuples with cross-assigned projection matrices / dtrs and reversed pairs (E3'), the cost image
    # of evaluate(), and the correlation form (E6; 1 - cc cancels two numbers near 1: absolute tolerance)
    if n >= 3:
        k = int(rng.integers(1, 9))
        idx = np.stack([rng.integers(0, n, k), rng.integers(0, n, k), rng.integers(0, n, k), rng.integers(0, n, k)], 1)
        idx = idx[(idx[:, 0] != idx[:, 1])].astype(np.int32)
        far = [np.linalg.norm(Cs[a] - Cs[b]) >= 1e-4 * np.linalg.norm(Cs[a]) for a, b in idx[:, :2]]
        idx = idx[np.asarray(far, bool)] if len(idx) else idx
        if len(idx):
            out = np.zeros(len(idx), np.float32)
            m.evaluate(idx, out)
            w = oracle.evaluate_pairs(Ps, dtrs_h, n_u, n_v, idx, object_radius_mm=radius, dkappa=dkappa, is_derivative=derivative)
            fin = np.isfinite(w["pairs"])
            sc = np.maximum(np.abs(w["pairs"][fin]), 1e-3 * np.abs(ref).max() if ref.size else 1.0)
            r_idx = float(np.max(np.abs(out[fin] - w["pairs"][fin]) / sc)) if fin.any() else 0.0
            if r_idx > 2e-3:
                bad += 1
                flag += "  <-- INDEX LIST OUT OF TOLERANCE (%.2e)" % r_idx
        cost = np.full((n, n), -7.0, np.float32)
        m.evaluate(cost)
        iu = np.triu_indices(n, 1)
        if not (np.array_equal(cost[iu[1], iu[0]], vals, equal_nan=True) and np.all(cost[iu] == -7.0)):
            bad += 1
            flag += "  <-- COST IMAGE DIFFERS FROM THE PAIR VALUES"
        if ok_pairs.all() and rng.integers(0, 3) == 0:
            m.useCorrelation(True)
            oracle.set_use_corr(1)
            try:
                wc = oracle.evaluate_all(Ps, dtrs_h, n_u, n_v, object_radius_mm=radius, dkappa=dkappa, is_derivative=derivative)
            finally:
                oracle.set_use_corr(0)
            _, vc = m.evaluate_range(0, n_pairs, want_pairs=True)
            m.useCorrelation(False)
            d_cc = float(np.nanmax(np.abs(vc - wc["pairs"]) / np.maximum(np.abs(wc["pairs"]), 1e-2)))
            if not d_cc < 2e-3:
                bad += 1
                flag += "  <-- CORRELATION FORM OFF BY %.2e" % d_cc
    worst_mean, worst_pair = max(worst_mean, rel_mean), max(worst_pair, rel_pair)
    print("case %2d: n=%2d %3dx%3d bins %3dx%3d kind %d span %3.0f deriv %d r=%5.1f dk=%.3f | mean %.2e pair %.2e | "
          "degrees %s%s" % (c, n, n_u, n_v, n_alpha, n_t, kind, span, derivative, radius, dkappa, rel_mean, rel_pair,
                          {d: degs.count(d) for d in sorted(set(degs))}, flag), flush=True)
    m.close()
    for d in dtrs:
        d.close()
print("polynomial path: worst mean %.2e, worst pair %.2e; default mode: worst of mean / pair %.2e; %d of %d cases out of "
      "tolerance, %.1f s" % (worst_mean, worst_pair, worst_auto, bad, cases, time.time() - t_start))
sys.exit(1 if bad else 0)
