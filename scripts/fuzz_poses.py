#!/usr/bin/env python3
"""Randomised sweep of the batched pose evaluation (GPU box): python scripts/fuzz_poses.py [cases] [seed]
csrc/ecc_poses.hip against the sequential evaluations of the SAME library (ecc_metric_set_pose_batching(0): one
setProjectionMatrices + evaluate per pose) -- the contract is bit identity, so no tolerance: random numbers of views (2 ... 300:
both forms of the all-pairs sum, every tail length), bin grids, sampling modes, object radius / dkappa / use_corr, 1 ... 70
poses of 0 ... 40 moved views each (view 0, the last view, duplicates of one view across poses, poses equal to the base),
the delta form, the full-matrix form with random first / stride, a base the metric has or has not seen, and ordinary
evaluations, index lists and parameter changes between the batches.  ref for the pattern: Gui/Visualization.h:59-112."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import epipolarconsistency_amd as E  # noqa: E402
from epipolarconsistency_amd import geometry, synthetic  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = np.random.default_rng(seed)
ctx = E.Context(0)
bad = 0
t0 = time.time()


def eq(x, y):
    return x == y or (np.isnan(x) and np.isnan(y))


for c in range(cases):
    n = int(rng.choice([2, 3, 4, 5, 9, 17, 33, 34, 64, 91, 130, 181, 257, 258, 300], p=[.05, .05, .05, .05, .1, .1, .1, .1, .1, .08, .06, .05, .04, .04, .03]))
    S = int(rng.choice([64, 96, 128]))
    Ba, Bt = int(rng.choice([32, 48, 64])), int(rng.choice([32, 48, 80]))
    Ps = synthetic.short_scan(n, S, S, 0.308 * 1024 / S, span_deg=float(rng.choice([200.0, 120.0, 360.0])))
    if rng.integers(0, 2):
        Ps = [P @ geometry.rigid_transform(*(rng.normal(0, 1.0, 3)), *(rng.normal(0, 0.01, 3))) for P in Ps]
    pool = [E.RadonIntermediate.from_host(ctx, rng.standard_normal((Bt, Ba)).astype(np.float32), S, S) for _ in range(min(n, 6))]
    dtrs = [pool[v % len(pool)] for v in range(n)]
    mode = str(rng.choice(["auto", "polynomial", "per_sample", "reference"], p=[.4, .3, .2, .1]))
    if mode == "reference" and n > 64:
        mode = "auto"
    radius = float(rng.choice([0.0, 0.0, 60.0, 110.0]))
    dkappa = float(rng.choice([0.0, 0.0, 0.0, 0.006]))
    corr = bool(rng.integers(0, 5) == 0)
    a = E.MetricRadonIntermediate(ctx, Ps, dtrs).setSampling(mode)
    b = E.MetricRadonIntermediate(ctx, Ps, dtrs).setSampling(mode).setPoseBatching(False)
    for m in (a, b):
        m.setObjectRadius(radius)
        m.setEpipolarPlaneStep(dkappa)
        m.useCorrelation(corr)
        if rng.integers(0, 4) == 0:
            m.setIncremental(True) if m is a else None
    P0 = E.pack_projection_matrices(Ps)
    ok = True
    why = ""
    for rnd in range(int(rng.integers(1, 4))):
        K = int(rng.integers(1, 71 if n <= 130 else 21))
        views, rows, poses = [], [], []
        for k in range(K):
            kind = rng.integers(0, 10)
            if kind == 0:
                vk = []
            elif kind == 1 and n > 2:
                vk = sorted(set(int(v) for v in rng.integers(0, n, size=int(rng.integers(2, min(n, 41))))))
            elif kind == 2:
                vk = [0]
            elif kind == 3:
                vk = [n - 1]
            else:
                vk = [int(rng.integers(0, n))]
            P = P0.copy()
            for v in vk:
                T = geometry.rigid_transform(*(rng.normal(0, 0.8, 3)), *(rng.normal(0, 0.004, 3)))
                P[v] = (P0[v].reshape(4, 3).T @ T).T.reshape(12)
            views.append(vk)
            rows.append(P[vk].copy() if vk else np.zeros((0, 12)))
            poses.append(P)
        want = np.array([b.setProjectionMatrices(P).evaluate() for P in poses])
        form = int(rng.integers(0, 3))
        if form == 0:
            a.setProjectionMatrices(P0)
            got = a.evaluate_pose_deltas(views, rows)
            tail = (a.evaluate(), b.setProjectionMatrices(P0).evaluate())
            same = np.array_equal(got, want, equal_nan=True) and eq(*tail)
        elif form == 1:
            if rng.integers(0, 2):
                a.setProjectionMatrices(P0)  # else: whatever the metric holds (the first pose becomes the base if that is far)
            got = a.evaluate_poses(poses)
            tail = (a.evaluate(), b.setProjectionMatrices(poses[-1]).evaluate())
            same = np.array_equal(got, want, equal_nan=True) and eq(*tail)
        else:
            first, stride = int(rng.integers(0, 3)), int(rng.integers(1, 5))
            got = a.evaluate_poses(poses, first=first, stride=stride)
            sel = np.arange(first, K, stride)
            tail = (0.0, 0.0)
            same = np.array_equal(got[sel], want[sel], equal_nan=True) and not np.delete(got, sel).any()
        if not same:
            ok = False
            sel_ = np.arange(K) if form != 2 else sel
            d = [int(q) for q in sel_ if not (got[q] == want[q] or (np.isnan(got[q]) and np.isnan(want[q])))]
            why = "round %d form %d K %d batched %d; poses %s moved %s got %s want %s; evaluate afterwards %r" % (
                rnd, form, K, a.last_batched_poses(), d[:4], [views[q] for q in d[:4]], [float(got[q]) for q in d[:4]], [float(want[q]) for q in d[:4]], tail)
            break
        # something else in between
        other = int(rng.integers(0, 4))
        if other == 0:
            x, y = a.evaluate(), b.setProjectionMatrices(a._Ps).evaluate()
            if not (x == y or (np.isnan(x) and np.isnan(y))):
                ok, why = False, "evaluate after batch: %r vs %r" % (x, y)
        elif other == 1 and n >= 3:
            idx = rng.integers(0, n, size=(int(rng.integers(1, 30)), 2)).astype(np.int32)
            idx = np.ascontiguousarray(np.concatenate([idx, idx], axis=1))
            b.setProjectionMatrices(a._Ps)
            x, y = a.evaluate(idx), b.evaluate(idx)
            if not (x == y or (np.isnan(x) and np.isnan(y))):
                ok, why = False, "index list after batch: %r vs %r" % (x, y)
        elif other == 2:
            radius2 = float(rng.choice([0.0, 75.0]))
            a.setObjectRadius(radius2)
            b.setObjectRadius(radius2)
        if not ok:
            break
    bad += 0 if ok else 1
    print("case %3d: n=%3d %3d^2 bins %dx%d %-10s r=%5.1f dk=%.3f corr %d: %s %s" % (c, n, S, Ba, Bt, mode, radius, dkappa, corr,
                                                                                  "ok" if ok else "MISMATCH", why), flush=True)
    a.close()
    b.close()
    for d in pool:
        d.close()
print("%d of %d cases differ, %.1f s" % (bad, cases, time.time() - t0))
sys.exit(1 if bad else 0)
