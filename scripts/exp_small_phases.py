"""Phase time-line of small_eval_kernel (experiments; needs a library built with -DECC_SMALL_STAMPS -- python -c "from
epipolarconsistency_amd import build; build.build_library(force=True, extra_flags=['-DECC_SMALL_STAMPS'])";
the stamps cost registers (occupancy 3 instead of 5), so launches of more than 768 workgroups run in two rounds here): wall-clock stamps (100 MHz) per workgroup --
start, records done (phase A), value stored (phase B), and for the last arriver the sum stored."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
mode = sys.argv[3] if len(sys.argv) > 3 else "auto"
use_list = len(sys.argv) > 4  # 5th argument: evaluate the n - 1 pairs of view n / 2 as an index list instead of all pairs
B = 768
dev = torch.device("cuda", 0)
Ps = synthetic.short_scan(n, S, S, 0.308 * 1024 / S)
ctx = E.Context(0)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
dtrs = []
for a in range(0, n, 50):
    imgs = synthetic.projections_torch(Ps[a:a + 50], S, S, synthetic.sphere_phantom(), dev)
    dtrs += E.RadonIntermediate.compute_into(ctx, imgs, slabs[a:a + 50], B, B)
    ctx.synchronize()
m = E.MetricRadonIntermediate(ctx, Ps, dtrs).setSampling(mode)
P = E.pack_projection_matrices(Ps)
n_pairs = n * (n - 1) // 2
idx4 = None
if use_list:
    idx4 = np.array([(min(n // 2, v), max(n // 2, v), min(n // 2, v), max(n // 2, v)) for v in range(n) if v != n // 2], np.int32)
    n_pairs = len(idx4)
    vals = np.empty(n_pairs, np.float32)
wpp = 4 if n_pairs <= 1024 else (2 if n_pairs <= 2048 else 1)
blocks = (n_pairs + 4 // wpp - 1) // (4 // wpp)
L = _lib.lib()
L.ecc_debug_small_stamps.argtypes = [C.c_void_p, C.c_int]
for rep in range(6):
    m.setProjectionMatrices(P)
    if use_list:
        m.evaluate(idx4, vals)
    else:
        m.evaluate()
    st = np.zeros((blocks, 4), np.uint64)
    assert L.ecc_debug_small_stamps(C.c_void_p(st.ctypes.data), blocks) == 0
    t = st.astype(np.int64)
    t0 = t[:, 0].min()
    last = int(np.argmax(t[:, 3]))
    print("rep %d: %d workgroups (%d waves per pair); starts spread %.2f us; phase A median %.2f max %.2f; phase B median %.2f max %.2f; "
          "last value stored at %.2f us; sum stored at %.2f us (workgroup %d)"
          % (rep, blocks, wpp, (t[:, 0].max() - t0) / 100.0, np.median(t[:, 1] - t[:, 0]) / 100.0, (t[:, 1] - t[:, 0]).max() / 100.0,
             np.median(t[:, 2] - t[:, 1]) / 100.0, (t[:, 2] - t[:, 1]).max() / 100.0, (t[:, 2].max() - t0) / 100.0,
             0.0, last))
