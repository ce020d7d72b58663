"""Pair-kernel experiments on the BASELINE workload: kernel time (HIP events), step time and parity of the mean against
the oracle, for the library as currently built / configured (POLY_TOL in this script's environment -> MetricRadonIntermediate.debugSetPolyTolerance).  python scripts/exp_pairs.py [tag]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import geometry, synthetic
n, S, B = 400, 1024, 768
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(dev))
Ps = synthetic.short_scan(n, S, S, 0.308)
ph = synthetic.sphere_phantom()
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
dtrs = []
for a in range(0, n, 50):
    imgs = synthetic.projections_torch(Ps[a:a + 50], S, S, ph, dev)
    dtrs += E.RadonIntermediate.compute_into(ctx, imgs, slabs[a:a + 50], B, B)
    ctx.synchronize()
m = E.MetricRadonIntermediate(ctx, Ps, dtrs)
if os.environ.get("POLY_TOL"):
    m.debugSetPolyTolerance(float(os.environ["POLY_TOL"]))
P = E.pack_projection_matrices(Ps)
for _ in range(100):
    m.setProjectionMatrices(P).evaluate()
ctx.enable_timing(True)
ks = []
for _ in range(50):
    m.setProjectionMatrices(P).evaluate()
    ks.append(ctx.last_kernel_ms("pairs"))
ctx.enable_timing(False)
ts = []
for b in range(7):
    t0 = time.perf_counter()
    for _ in range(100):
        m.setProjectionMatrices(P).evaluate()
    ts.append((time.perf_counter() - t0) / 100)
val = m.evaluate()
degs = np.bincount([p["degree"] for p in m.debug_polynomials(0, 20000)], minlength=11).tolist()
out = dict(tag=sys.argv[1] if len(sys.argv) > 1 else "", kernel_ms_median=float(np.median(ks)), kernel_ms_min=float(np.min(ks)),
           step_ms_median=1e3 * float(np.median(ts)), value=val, degree_histogram_first_20000=degs)
ref_path = "/tmp/ecc_ref_mean.json"
if os.path.exists(ref_path):
    ref = json.load(open(ref_path))["mean"]
else:
    import oracle
    oracle.build(native=True)
    host = [d.readback() for d in dtrs]
    ref = oracle.evaluate_all(Ps, host, S, S, native=True)["mean"]
    json.dump({"mean": ref}, open(ref_path, "w"))
out["rel_err_vs_oracle"] = abs(val - ref) / ref
print(json.dumps(out))
