"""Diagnostic (GPU box): per-pair parity of both kernel variants against the oracle at a chosen size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic
import oracle

n, S, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
Ps = synthetic.short_scan(n, S, S, 0.308 * 1024 / S)
ph = synthetic.sphere_phantom()
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
dtrs = []
for a in range(0, n, 50):
    imgs = synthetic.projections_torch(Ps[a:a + 50], S, S, ph, dev)
    dtrs += E.RadonIntermediate.compute_into(ctx, imgs, slabs[a:a + 50], B, B)
    ctx.synchronize()
m = E.MetricRadonIntermediate(ctx, Ps, dtrs)
n_pairs = n * (n - 1) // 2
res = {0: m.evaluate_range(0, n_pairs, want_pairs=True)}
s0, v0 = res[0]
host = [d.readback() for d in dtrs]
oracle.build(native=True)
ref = oracle.evaluate_all(Ps, host, S, S, native=True)
L = oracle.lib(True)
L.eccor_set_variant(1)
hi = oracle.evaluate_all(Ps, host, S, S, native=True)
L.eccor_set_variant(2)
lm = oracle.evaluate_all(Ps, host, S, S, native=True)
probes = {}
L.eccor_set_variant(1)
for bits in (1, 2, 4, 8, 16, 32, 64, 17, 81, 127):
    L.eccor_set_probe(bits)
    probes[bits] = oracle.evaluate_all(Ps, host, S, S, native=True)
L.eccor_set_probe(0)
L.eccor_set_variant(0)
r = ref["pairs"].astype(np.float64)
def rep(name, v, mean):
    v = v.astype(np.float64)
    rel = np.abs(v - r) / np.abs(r)
    print("%-22s mean rel %.3e | pair rel median %.2e p99 %.2e max %.2e | signed mean rel of pairs %.3e | sum|d|/sum %.3e" % (
        name, abs(mean - ref["mean"]) / ref["mean"], np.median(rel), np.quantile(rel, 0.99), rel.max(),
        np.mean((v - r) / r), np.abs(v - r).sum() / r.sum()))
for var in res:
    rep("gpu variant %d" % var, res[var][1], res[var][0] / n_pairs)
rep("oracle f64 geometry", hi["pairs"], hi["mean"])
rep("oracle glibc floats", lm["pairs"], lm["mean"])
for bits in probes:
    rep("oracle f64 probe %d" % bits, probes[bits]["pairs"], probes[bits]["mean"])
big = np.argsort(-r)[:10]
print("largest pairs", [(E.get_ij(int(b), n), float(r[b])) for b in big[:5]], "share of top 1%:", np.sort(r)[-n_pairs // 100:].sum() / r.sum())
d = (v0.astype(np.float64) - r)
worst = np.argsort(-np.abs(d))[:8]
print("worst abs contributions (fast):", [(E.get_ij(int(b), n), float(d[b] / r.sum()), float(d[b] / r[b])) for b in worst])
