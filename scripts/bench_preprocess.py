"""Pre-processing micro-bench (GPU box): us per 1024^2 image, reference defaults + cosine weight, HIP-event timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda", 0)
Ps = synthetic.short_scan(400, S, S, 0.308 * 1024 / S)[:n]
imgs = torch.rand((n, S, S), dtype=torch.float32, device=dev)
out = torch.empty_like(imgs)
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.enable_timing(True)
for name, setup in (("defaults+cos", lambda p: None), ("no lowpass", lambda p: setattr(p.lowpass, "gaussian_sigma", 0.0)),
                    ("log+flip", lambda p: (setattr(p.intensity, "apply_log", True), setattr(p.image_geometry, "flip_u", True))),
                    ("normalize", lambda p: setattr(p.intensity, "normalize", True))):
    pp = E.PreProccess()
    setup(pp)
    for r in range(3):
        pp.process(ctx, imgs, Ps, out=out)
    us = 1e3 * ctx.last_kernel_ms("preprocess") / n
    print("%-14s %.2f us per image, %.0f GB/s of 8 B/pixel" % (name, us, 8.0 * S * S / (us * 1e-6) / 1e9))
# calibration of the byte counters on this shape (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated"): a plain
# copy of the same stack -- 4 B read and 4 B written per pixel, nothing else
for r in range(3):
    out.copy_(imgs)
torch.cuda.synchronize()
