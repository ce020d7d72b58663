"""Experiment: what do the kappa_max = pi/2 pairs (baseline through the object, 3.5 % of the BASELINE workload's pairs) cost
INSIDE the mixed launch, and what do the row-quad copies (Context.debugSetQuadCopies) return there?  Per copy layout, kernel
time by HIP events of
  (a) the product's all-pairs launch,
  (b) an index list over all 79 800 pairs in get_ij order (same kernel, the list form),
  (c) that list without the heavy pairs (less work AND no heavy waves),
  (d) that list with every heavy pair REPLACED by a light pair of the same row of the pair triangle (same amount of ordinary work,
      no heavy waves): (b) - (d) is what the heavy waves cost beyond an ordinary pair's share, (d) - (c) the ordinary share.
python scripts/exp_heavy_gap.py"""
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
N = n * (n - 1) // 2
iu = np.triu_indices(n, 1)
out = {}


def timed(m, idx4=None, reps=17):
    vals = None if idx4 is None else np.empty(len(idx4), np.float32)
    ctx.enable_timing(True)
    ks = []
    for _ in range(reps):
        if idx4 is None:
            m.evaluate()
        else:
            m.evaluate(idx4, vals)
        ks.append(ctx.last_kernel_ms("pairs"))
    ctx.enable_timing(False)
    return 1e3 * float(np.median(ks[3:]))


for quads in (0, 1, 0, 1):
    ctx.debugSetQuadCopies(bool(quads))
    m = E.MetricRadonIntermediate(ctx, Ps, dtrs).setSampling("polynomial")
    # the pairs on the per-sample path (fit rejected: kappa_max = pi/2, the sampling curve crosses the whole Radon intermediate)
    heavy = np.concatenate([[p["degree"] for p in m.debug_polynomials(a, min(10000, N - a))] for a in range(0, N, 10000)]) == 0
    nat = np.arange(N)
    # replacement: the pair 100 places earlier in the same row (|i - j| smaller by 100: kappa_max well below pi/2)
    repl = nat.copy()
    repl[heavy] = nat[heavy] - 100
    d = iu[1] - iu[0]
    sys.stderr.write("heavy %d, |i-j| of heavy pairs %d..%d, heavy after replacement %d, row changes %d\n"
                     % (heavy.sum(), d[heavy].min(), d[heavy].max(), heavy[repl].sum(), (iu[0][repl] != iu[0][nat]).sum()))
    assert not heavy[repl].any() and (iu[0][repl] == iu[0][nat]).all()

    def lst(o):
        return np.stack([iu[0][o], iu[1][o], iu[0][o], iu[1][o]], 1).astype(np.int32)

    r = dict(heavy_pairs=int(heavy.sum()),
             all_pairs_launch_us=timed(m),
             list_all_us=timed(m, lst(nat)),
             list_without_heavy_us=timed(m, lst(nat[~heavy])),
             list_heavy_replaced_us=timed(m, lst(repl)))
    r["heavy_extra_us"] = r["list_all_us"] - r["list_heavy_replaced_us"]
    r["ordinary_share_us"] = r["list_heavy_replaced_us"] - r["list_without_heavy_us"]
    out.setdefault("quads" if quads else "paired", []).append(r)
    m.close()
print(json.dumps(out, indent=1))
