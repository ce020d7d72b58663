"""Radon kernel statistics (GPU box): builds with -DECC_RADON_STATS, prints slab counts, mean slab shape and how many
steps took the LDS path / the global-memory path inside slabs / the safety net."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epipolarconsistency_amd import build
build.build_library(force=True, extra_flags=["-DECC_RADON_STATS"])
import torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic, _lib
dev = torch.device("cuda", 0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 768
n = 2
Ps = synthetic.short_scan(400, S, S, 0.308 * 1024 / S)[100:100 + n]
imgs = synthetic.projections_torch(Ps, S, S, synthetic.sphere_phantom(), dev)
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
L = _lib.lib()
out = (C.c_ulonglong * 8)()
L.ecc_debug_radon_stats(out, 1)
keep = E.RadonIntermediate.compute_into(ctx, imgs, slabs, B, B); ctx.synchronize()
L.ecc_debug_radon_stats(out, 1)
v = list(out)
print("per image: slabs %.0f (mean stride %.1f pairs, %.1f rows), steps: LDS %.4g, global inside slabs %.4g, safety net %.4g" % (
    v[0] / n, v[4] / max(v[0], 1), v[5] / max(v[0], 1), v[1] / n, v[2] / n, v[3] / n))
