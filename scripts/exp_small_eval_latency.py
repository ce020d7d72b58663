"""Latency of small evaluations (index lists as in Gui/SingleImageMotion.h's n-1 pairs of one view, or a single pair) per
sampling mode on the BASELINE data set: what ECC_SAMPLING_AUTO's choice of the reference arithmetic costs.
usage: exp_small_eval_latency.py [modes, comma separated] [list: 1|399|512]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic
n, S, B = 400, 1024, 768
dev = torch.device("cuda", 0)
Ps = synthetic.short_scan(n, S, S, 0.308)
ph = synthetic.sphere_phantom()
torch.cuda.set_stream(torch.cuda.Stream(dev))  # a stream of our own, not the legacy default stream
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
dtrs = []
for a in range(0, n, 50):
    imgs = synthetic.projections_torch(Ps[a:a + 50], S, S, ph, dev)
    dtrs += E.RadonIntermediate.compute_into(ctx, imgs, slabs[a:a + 50], B, B)
    ctx.synchronize()
m = E.MetricRadonIntermediate(ctx, Ps, dtrs)
P = E.pack_projection_matrices(Ps)
MODES = sys.argv[1].split(",") if len(sys.argv) > 1 else ("polynomial", "per_sample", "reference")
ONLY = sys.argv[2] if len(sys.argv) > 2 else None
out = {}
for name, idx in (("1 pair", [(10, 250)]), ("399 pairs of view 200", [(min(200, v), max(200, v)) for v in range(n) if v != 200]),
                  ("512 random pairs", None)):
    if ONLY and not name.startswith(ONLY):
        continue
    if idx is None:
        rng = np.random.default_rng(0)
        idx = [tuple(sorted(rng.choice(n, 2, replace=False))) for _ in range(512)]
    idx4 = np.array([(a, b, a, b) for a, b in idx], np.int32)
    vals = np.empty(len(idx4), np.float32)
    row = {}
    for mode in MODES:
        m.setSampling(mode)
        for _ in range(20):
            m.setProjectionMatrices(P); m.evaluate(idx4, vals)
        t0 = time.perf_counter()
        for _ in range(200):
            m.setProjectionMatrices(P); m.evaluate(idx4, vals)
        row[mode + "_us"] = 1e6 * (time.perf_counter() - t0) / 200
    out[name] = row
print(json.dumps(out))
