"""Diagnostic (library built with -DPK_EXP_STAMPS: scripts/experiments/build_variant.sh stamps pairs_kernel.hip -DPK_EXP_STAMPS,
run with ECC_HIP_LIB=scripts/experiments/_build/libecc_stamps.so): start / end time, XCC and CU of every pair's wave in one
all-pairs launch of the BASELINE workload -> where the kernel's time goes by pair class, when each XCD finishes, and what a
kappa_max = pi/2 wave does to the waves that share its CU.
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
N = n * (n - 1) // 2
m = E.MetricRadonIntermediate(ctx, Ps, dtrs).setSampling("polynomial")
if hasattr(m, "debugSetXcdSchedule"):  # only with xcd_schedule_table.patch applied
    m.debugSetXcdSchedule(int(os.environ.get("XCD_SCHEDULE", "1")))
for _ in range(10):
    m.evaluate()
ctx.enable_timing(True)
ks = []
for _ in range(8):
    m.evaluate()
    ks.append(ctx.last_kernel_ms("pairs"))
ctx.enable_timing(False)
import time
t_s = float(os.environ.get("SUSTAIN_S", "0"))  # evaluations back to back for this long first: the socket at its power cap
t0 = time.perf_counter()
while time.perf_counter() - t0 < t_s:
    m.evaluate()
order = ""
K = m.debug_K01(0, N)  # its own all-pairs launch, stamped (the first one warms the debug buffers)
for _ in range(int(200 * min(t_s, 1.0))):
    m.evaluate()
K = m.debug_K01(0, N)
raw = K.view(np.uint32)
if os.environ.get("STAMPS_OUT"):
    np.savez_compressed(os.environ["STAMPS_OUT"], raw=raw[:, :6].copy())
t0 = raw[:, 0].astype(np.uint64) | (raw[:, 1].astype(np.uint64) << 32)
t1 = raw[:, 2].astype(np.uint64) | (raw[:, 3].astype(np.uint64) << 32)
xcc = raw[:, 4] & 0xf
hw = raw[:, 4] >> 4
cu = (hw >> 8) & 0xf
se = (hw >> 13) & 0x7
sh = (hw >> 12) & 1
simd = (hw >> 4) & 3
cuid = (xcc.astype(np.int64) << 8) | (se.astype(np.int64) << 5) | (sh.astype(np.int64) << 4) | cu  # one id per CU
cls = raw[:, 5] & 0xfe
base = t0.min()
s = (t0 - base).astype(np.float64) * 0.01  # us (100 MHz)
e = (t1 - base).astype(np.float64) * 0.01
dur = e - s
out = {"xcd_schedule": int(os.environ.get("XCD_SCHEDULE", "1")) if hasattr(m, "debugSetXcdSchedule") else None, "kernel_us_by_events": 1e3 * float(np.median(ks[3:])), "kernel_span_us": float(e.max()),
       "last_start_us": float(s.max()), "distinct_cus": int(len(np.unique(cuid)))}
for name, sel in (("exact_heavy", cls == 0), ("deg6", cls == 6), ("deg8", cls == 8), ("deg10", cls == 10)):
    if sel.sum():
        out[name] = dict(n=int(sel.sum()), mean_dur_us=float(dur[sel].mean()), p5=float(np.percentile(dur[sel], 5)),
                         p95=float(np.percentile(dur[sel], 95)), last_end_us=float(e[sel].max()), last_start_us=float(s[sel].max()))
bins = np.arange(0, e.max() + 10, 10.0)
out["running_waves_per_10us"] = [int(((s < b + 10) & (e > b)).sum()) for b in bins]
out["running_heavy_per_10us"] = [int(((s < b + 10) & (e > b) & (cls == 0)).sum()) for b in bins]
out["per_xcc_end_us"] = [float(e[xcc == x].max()) if (xcc == x).any() else None for x in range(8)]
out["per_xcc_heavy"] = [int(((xcc == x) & (cls == 0)).sum()) for x in range(8)]
out["per_xcc_waves"] = [int((xcc == x).sum()) for x in range(8)]
out["per_xcc_slot_us"] = [float(dur[xcc == x].sum()) for x in range(8)]
# what a heavy wave does to its neighbours: degree-8 waves by the number of heavy waves that overlap them on the same CU
d8 = np.flatnonzero(cls == 8)
hv = np.flatnonzero(cls == 0)
by_cu = {}
for i in hv:
    by_cu.setdefault(int(cuid[i]), []).append(i)
cnt = np.zeros(len(d8), np.int32)
ovl = np.zeros(len(d8))
for q, i in enumerate(d8):
    for j in by_cu.get(int(cuid[i]), ()):
        o_ = min(e[i], e[j]) - max(s[i], s[j])
        if o_ > 0:
            cnt[q] += 1
            ovl[q] += o_
out["deg8_dur_by_heavy_neighbours_on_cu"] = {str(k): dict(n=int((cnt == k).sum()), mean_dur_us=float(dur[d8][cnt == k].mean()))
                                              for k in range(0, 6) if (cnt == k).any()}
# the same per XCC: degree-8 waves by the number of heavy waves running in their XCC at their mid time
mid = 0.5 * (s[d8] + e[d8])
hx = np.zeros(len(d8), np.int32)
for x in range(8):
    hs, he = s[hv][xcc[hv] == x], e[hv][xcc[hv] == x]
    sel = xcc[d8] == x
    hx[sel] = [int(((hs <= t) & (he >= t)).sum()) for t in mid[sel]]
qs = [0, 1, 5, 10, 20, 40, 80, 1000]
out["deg8_dur_by_heavy_running_in_xcc"] = {"%d-%d" % (qs[k], qs[k + 1] - 1): dict(n=int(((hx >= qs[k]) & (hx < qs[k + 1])).sum()),
                                           mean_dur_us=float(dur[d8][(hx >= qs[k]) & (hx < qs[k + 1])].mean()))
                                           for k in range(len(qs) - 1) if ((hx >= qs[k]) & (hx < qs[k + 1])).any()}
print(json.dumps(out))
