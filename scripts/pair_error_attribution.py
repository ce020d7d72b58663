#!/usr/bin/env python3
"""Whose error is a single pair value's deviation from the oracle?  (VERDICT round 5, weak 1 ii.)
Config 2 (64 views of 512^2, 512^2 bins; `--views/--size/--bins` for others): all pair values of the library's sampling modes
against (a) the normative oracle -- the reference's fp32 line -> (angle, distance) mapping, ref: ...RadonIntermediate.cu:71-113,
EpipolarConsistencyCommon.hxx:152-171 -- and (b) its variant 1, the same mapping in binary64 rounded once (the noise-free
values of the same formula).  Prints max / p99 / p50 of the relative differences, and the means'."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import epipolarconsistency_amd as E  # noqa: E402
from epipolarconsistency_amd import synthetic  # noqa: E402
import oracle  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=64)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--bins", type=int, default=512)
a = ap.parse_args()
n, S, B = a.views, a.size, a.bins
ctx = E.Context(0)
Ps = synthetic.short_scan(n, S, S, 0.308 * 1024.0 / S)
dev = torch.device("cuda", 0)
imgs = synthetic.projections_torch(Ps, S, S, synthetic.sphere_phantom(), dev)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
dtrs = E.RadonIntermediate.compute_into(ctx, imgs, slabs, B, B)
ctx.synchronize()
host = [d.readback() for d in dtrs]
n_pairs = n * (n - 1) // 2
oracle.build(native=True)
ref = oracle.evaluate_all(Ps, host, S, S, native=True)
oracle.set_variant(1, native=True)
ref64 = oracle.evaluate_all(Ps, host, S, S, native=True)
oracle.set_variant(0, native=True)
p32, p64 = np.asarray(ref["pairs"], np.float64), np.asarray(ref64["pairs"], np.float64)


def dist(x, y):
    r = np.abs(x - y) / np.maximum(np.abs(y), 1e-300)
    return {"max": float(r.max()), "p99": float(np.percentile(r, 99)), "p50": float(np.percentile(r, 50))}


out = {"workload": "%d views of %d^2, %d^2 bins, %d pairs" % (n, S, B, n_pairs),
       "normative_oracle_vs_float64_geometry": dist(p32, p64),
       "mean_normative_vs_float64_geometry": abs(ref["mean"] - ref64["mean"]) / abs(ref64["mean"])}
m = E.MetricRadonIntermediate(ctx, Ps, dtrs)
for mode in ("polynomial", "per_sample", "reference"):
    if mode == "reference" and n_pairs > 5000:
        continue
    total, vals = m.setSampling(mode).evaluate_range(0, n_pairs, want_pairs=True)
    g = vals.astype(np.float64)
    out[mode] = {"vs_normative_oracle": dist(g, p32), "vs_float64_geometry": dist(g, p64),
                 "mean_vs_normative": abs(total / n_pairs - ref["mean"]) / abs(ref["mean"]),
                 "mean_vs_float64_geometry": abs(total / n_pairs - ref64["mean"]) / abs(ref64["mean"])}
print(json.dumps(out, indent=1))
