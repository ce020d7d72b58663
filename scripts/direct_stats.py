"""MetricDirect's line-integral kernel (GPU box): library built with -DECC_DIRECT_STATS (scripts/experiments/build_variant.sh dstats
direct_kernel.hip -DECC_DIRECT_STATS; run with ECC_HIP_LIB=.../_build/libecc_dstats.so) -- slabs per workgroup, mean slab shape, and how many
sampling steps ran through the LDS tile, through global memory inside the slab loop, and behind it.
python scripts/direct_stats.py [views] [size]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda", 0)
Ps = synthetic.short_scan(n, S, S, 0.308 * 1024 / S)
imgs = synthetic.projections_torch(Ps, S, S, synthetic.sphere_phantom(), dev)
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
m = E.MetricDirect(ctx, Ps, imgs)
L = C.CDLL(_lib.LIB_PATH)
out = (C.c_ulonglong * 8)()
m.evaluate()
L.ecc_debug_direct_stats(out, 1)
v = m.evaluate()
L.ecc_debug_direct_stats(out, 1)
st = list(out)
pairs = n * (n - 1) // 2
print("pairs %d, value %.6g" % (pairs, v))
print("slabs %d (mean S %.1f, mean H %.1f)" % (st[0], st[4] / max(st[0], 1), st[5] / max(st[0], 1)))
tot = st[1] + st[2] + st[3]
print("steps: tile %d (%.1f %%), global inside the slab loop %d (%.1f %%), global behind it %d (%.1f %%)"
      % (st[1], 100.0 * st[1] / tot, st[2], 100.0 * st[2] / tot, st[3], 100.0 * st[3] / tot))
