"""Where one optimiser step's wall time goes (GPU box): host-side timing of the two API calls vs kernel time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic

n, S, B = 400, 1024, 768
dev = torch.device("cuda", 0)
Ps = synthetic.short_scan(n, S, S, 0.308)
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
slabs = torch.randn((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
dtrs = [E.RadonIntermediate.wrap_device(ctx, slabs[k], B, B, S, S) for k in range(n)]
m = E.MetricRadonIntermediate(ctx, Ps, dtrs)
P = E.pack_projection_matrices(Ps)
for timing in (False, True):
    ctx.enable_timing(timing)
    for _ in range(50):
        m.setProjectionMatrices(P); m.evaluate()
    ts, te, tk = 0.0, 0.0, 0.0
    N = 300
    t0 = time.perf_counter()
    for _ in range(N):
        a = time.perf_counter()
        m.setProjectionMatrices(P)
        b = time.perf_counter()
        m.evaluate()
        c = time.perf_counter()
        ts += b - a; te += c - b
        if timing:
            tk += ctx.last_kernel_ms("pairs")
    tot = time.perf_counter() - t0
    print("timing events %s: step %.1f us = setProjectionMatrices %.1f us + evaluate %.1f us (k01+pairs kernels %.1f us)"
          % (timing, 1e6 * tot / N, 1e6 * ts / N, 1e6 * te / N, 1e3 * tk / N))
