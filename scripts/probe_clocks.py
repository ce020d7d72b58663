"""Samples the GPU's clock and power (rocm-smi) while the BASELINE evaluation runs back to back for a few seconds: is the pair
kernel running at the chip's top clock or power-limited?  python scripts/probe_clocks.py [seconds]"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
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
samples = []
stop = False


def sampler():
    while not stop:
        t = time.perf_counter()
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--csv"], capture_output=True, text=True, timeout=5).stdout
        except Exception as e:  # noqa
            o = "error %r" % (e,)
        samples.append((t, o))
        time.sleep(0.3)


print(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True).stdout)
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter(); k = 0; marks = []
ctx.enable_timing(True)
while time.perf_counter() - t0 < secs:
    for _ in range(100):
        m.setProjectionMatrices(P); m.evaluate()
    k += 100
    marks.append((time.perf_counter() - t0, ctx.last_kernel_ms("pairs")))
stop = True; th.join()
print("steps", k, "us per step", 1e6 * (time.perf_counter() - t0) / k)
print("kernel ms over time:", [(round(a, 2), round(b, 4)) for a, b in marks[:: max(1, len(marks) // 12)]])
for t, o in samples:
    print(round(t - t0, 2), o.replace("\n", " | ")[:600])
