import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epipolarconsistency_amd import build
build.build_library(force=True, extra_flags=["-DECC_RADON_STATS"])
import torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic, _lib
dev = torch.device("cuda", 0)
S, B, n = 1024, 768, 2
Ps = synthetic.short_scan(400, S, S, 0.308)[100:100 + n]
imgs = synthetic.projections_torch(Ps, S, S, synthetic.sphere_phantom(), dev)
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
L = _lib.lib()
out = (C.c_ulonglong * 8)()
L.ecc_debug_radon_stats(out, 1)
keep = E.RadonIntermediate.compute_into(ctx, imgs, slabs, B, B); ctx.synchronize()
L.ecc_debug_radon_stats(out, 1)
v = list(out)
print("per image: chunks %.0f, fits %.0f, any %.0f, global-path steps %.3g, lds-path steps %.3g, mean w %.1f h %.1f" % (
    v[0] / n, v[1] / n, v[2] / n, v[3] / n, v[4] / n, v[5] / max(v[2], 1), v[6] / max(v[2], 1)))
