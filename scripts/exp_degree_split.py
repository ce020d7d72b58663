"""Per-polynomial degrees after economisation on the BASELINE scan: would loops specialised on (angle degree, distance degree)
instead of the maximum over a pair's four polynomials save Horner steps?"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic
n, S, B = 400, 1024, 768
ctx = E.Context(0)
Ps = synthetic.short_scan(n, S, S, 0.308)
slab = torch.zeros((1, E.slab_floats(B, B)), dtype=torch.float32, device="cuda")
d = E.RadonIntermediate.wrap_device(ctx, slab[0], B, B, S, S)
m = E.MetricRadonIntermediate(ctx, Ps, [d] * n).setSampling("polynomial")
m.evaluate()
hist = collections.Counter()
def deg(c):
    nz = np.nonzero(c[:11])[0]
    d = int(nz.max()) if len(nz) else 0
    return d + (d & 1)  # even degree class
for a in range(0, 79800, 10000):
    for r in m.debug_polynomials(a, min(10000, 79800 - a)):
        if not r["poly_ok"]:
            hist["exact"] += 1
            continue
        da = max(deg(r["ca"][0]), deg(r["ca"][1])); dd = max(deg(r["cd"][0]), deg(r["cd"][1]))
        hist[(r["degree"], da, dd)] += 1
for k, v in sorted(hist.items(), key=lambda kv: -kv[1]):
    print(k, v)
