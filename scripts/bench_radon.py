"""Radon-intermediate micro-bench (GPU box): ms per 1024^2 -> 768^2 derivative dtr, HIP-event timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
B = int(sys.argv[3]) if len(sys.argv) > 3 else 768
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
modes = sys.argv[5].split(",") if len(sys.argv) > 5 else ["exact", "fma", "exact", "fma"]  # A/B/A/B on one box
dev = torch.device("cuda", 0)
Ps = synthetic.short_scan(400, S, S, 0.308 * 1024 / S)[:n]
imgs = synthetic.projections_torch(Ps, S, S, synthetic.sphere_phantom(), dev)
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.enable_timing(True)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
for mode in modes:
    ctx.setRadonArithmetic(mode)
    for r in range(reps):
        keep = E.RadonIntermediate.compute_into(ctx, imgs, slabs, B, B)
        ctx.synchronize()
        print("%s rep %d: %.3f ms per Radon intermediate (%d images)" % (mode, r, ctx.last_kernel_ms("radon") / n, n))
