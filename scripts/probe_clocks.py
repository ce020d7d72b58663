"""How much of the pair kernel's time follows the engine clock?  The BASELINE step (one moved view + all 79 800 pairs) runs for
`secs` seconds per phase with an idle gap of 0 / 0.3 / 1 / 3 ms between steps: back to back the socket sits at its 1400-W cap and
the engine clock is throttled, with gaps the average power drops below the cap and the clock returns to its top.  Per phase: pair
kernel time by HIP events (median of the second half), socket power and engine clock from hwmon (mean of the second half).
python scripts/probe_clocks.py [secs]"""
import glob, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
n, S, B = 400, 1024, 768
dev = torch.device("cuda", 0)
Ps = synthetic.short_scan(n, S, S, 0.308)
ph = synthetic.sphere_phantom()
torch.cuda.set_stream(torch.cuda.Stream(dev))
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
for a in range(0, n, 50):
    imgs = synthetic.projections_torch(Ps[a:a + 50], S, S, ph, dev)
    keep = E.RadonIntermediate.compute_into(ctx, imgs, slabs[a:a + 50], B, B)
    ctx.synchronize()
dtrs = [E.RadonIntermediate.wrap_device(ctx, slabs[k], B, B, S, S) for k in range(n)]
m = E.MetricRadonIntermediate(ctx, Ps, dtrs)
P = E.pack_projection_matrices(Ps)
P2 = P.copy()
P2.reshape(-1)[200 * 12 + 9] += 1e-3
pr = torch.cuda.get_device_properties(dev)
pci = "%04x:%02x" % (pr.pci_domain_id, pr.pci_bus_id)
cands = [d for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")) if os.path.exists(d + "/power1_input") and os.path.exists(d + "/freq1_input")]
hw = next((d for d in cands if pci in os.path.realpath(os.path.dirname(os.path.dirname(d)))), cands[0] if len(cands) == 1 else None)
out = []
for gap_ms in (0.0, 0.3, 1.0, 3.0, 0.0):
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            try:
                samples.append((time.perf_counter(), float(open(hw + "/power1_input").read()) * 1e-6, float(open(hw + "/freq1_input").read()) * 1e-6))
            except Exception:
                pass
            time.sleep(0.005)
    th = threading.Thread(target=sampler)
    if hw:
        th.start()
    ctx.enable_timing(True)
    ks, k = [], 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < secs:
        m.setProjectionMatrices(P2 if k & 1 else P); m.evaluate(); k += 1
        ks.append((time.perf_counter() - t0, ctx.last_kernel_ms("pairs")))
        if gap_ms:
            t1 = time.perf_counter()
            while time.perf_counter() - t1 < gap_ms * 1e-3:
                pass
    ctx.enable_timing(False)
    stop[0] = True
    if hw:
        th.join()
    late = [v for t, v in ks if t > 0.5 * secs]
    ps = [(p, f) for t, p, f in samples if t - t0 > 0.5 * secs]
    r = dict(gap_ms=gap_ms, steps=k, kernel_us=1e3 * float(np.median(late)), kernel_us_p10=1e3 * float(np.percentile(late, 10)),
             watts=float(np.mean([p for p, f in ps])) if ps else None, sclk_mhz=float(np.mean([f for p, f in ps])) if ps else None)
    out.append(r)
    sys.stderr.write(json.dumps(r) + "\n")
print(json.dumps(out, indent=1))
