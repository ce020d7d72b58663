"""Diagnostic (library built with -DPK_EXP_STAMPS): start / end time of every pair's wave in one all-pairs launch of the
BASELINE workload -> where the kernel's time goes by pair class and how long its tail is."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic
n, S, B = 400, 1024, 768
dev = torch.device("cuda", 0)
Ps = synthetic.short_scan(n, S, S, 0.308)
ph = synthetic.sphere_phantom()
ctx = E.Context(0)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
dtrs = []
for a in range(0, n, 50):
    imgs = synthetic.projections_torch(Ps[a:a + 50], S, S, ph, dev)
    dtrs += E.RadonIntermediate.compute_into(ctx, imgs, slabs[a:a + 50], B, B)
    ctx.synchronize()
m = E.MetricRadonIntermediate(ctx, Ps, dtrs).setSampling("polynomial")
for _ in range(20):
    m.evaluate()
K = m.debug_K01(0, 79800)
K = m.debug_K01(0, 79800)
raw = K.view(np.uint32)
t0 = raw[:, 0].astype(np.uint64) | (raw[:, 1].astype(np.uint64) << 32)
t1 = raw[:, 2].astype(np.uint64) | (raw[:, 3].astype(np.uint64) << 32)
xcc = raw[:, 4] & 0xf
cls = raw[:, 5]
base = t0.min()
s = (t0 - base).astype(np.float64) * 0.01  # us
e = (t1 - base).astype(np.float64) * 0.01
dur = e - s
out = {"kernel_span_us": float(e.max()), "last_start_us": float(s.max())}
for name, sel in (("exact_heavy", (cls & 0xff) == 0), ("deg8", (cls & 0xff) == 8), ("deg10", (cls & 0xff) == 10)):
    if sel.sum():
        out[name] = dict(n=int(sel.sum()), mean_dur_us=float(dur[sel].mean()), p95_dur_us=float(np.percentile(dur[sel], 95)),
                         last_end_us=float(e[sel].max()), last_start_us=float(s[sel].max()))
# occupancy over time: number of running waves in 10-us bins
bins = np.arange(0, e.max() + 10, 10.0)
occ = [(int(((s < b + 10) & (e > b)).sum())) for b in bins]
out["running_waves_per_10us"] = occ
out["per_xcc_end_us"] = [float(e[xcc == x].max()) if (xcc == x).any() else None for x in range(8)]
print(json.dumps(out))
