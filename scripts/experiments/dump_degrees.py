"""Degree class (0 = exact path, 4/6/8/10) of every pair of the BASELINE workload -> gpurun_out/degrees.npy"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic
n, S, B = 400, 1024, 768
dev = torch.device("cuda", 0)
Ps = synthetic.short_scan(n, S, S, 0.308)
ctx = E.Context(0)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
dtrs = [E.RadonIntermediate.wrap_device(ctx, slabs[k], B, B, S, S) for k in range(n)]
m = E.MetricRadonIntermediate(ctx, Ps, dtrs).setSampling("polynomial")
deg = np.concatenate([[p["degree"] for p in m.debug_polynomials(a, min(10000, 79800 - a))] for a in range(0, 79800, 10000)])
K = np.concatenate([m.debug_K01(a, min(10000, 79800 - a))[:, 15] for a in range(0, 79800, 10000)])
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/degrees.npy", deg.astype(np.int8))
np.save("gpurun_out/kappa_max.npy", K.astype(np.float32))
print(np.bincount(deg, minlength=11))
heavy = K > np.pi / 4
print("kappa_max > pi/4:", int(heavy.sum()), "pairs; degree histogram", np.bincount(deg[heavy], minlength=11))
print("others:", int((~heavy).sum()), "pairs; degree histogram", np.bincount(deg[~heavy], minlength=11))
