"""Soak of the one-launch path and the pinned-list path (GPU box): millions of small evaluations of changing sizes and kinds with
one view moved between calls, every result compared with the value the stream-ordered path gave for the same pose -- looks for
rare failures of the cross-workgroup hand-over (ticket, system-scope stores, the word the host polls).
usage: soak_small_eval.py [evaluations, default 1000000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic, geometry

total = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
S, B = 256, 128
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(dev))
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)
done, bad, t0 = 0, 0, time.time()
for n in (2, 5, 12, 20, 40):
    Ps = synthetic.short_scan(n, S, S, 0.308 * 1024 / S)
    dtrs = [E.RadonIntermediate.from_host(ctx, rng.standard_normal((B, B)).astype(np.float32), S, S) for _ in range(n)]
    P = E.pack_projection_matrices(Ps)
    poses = []
    for k in range(32):
        Pk = P.copy()
        Pk[n // 2] = (Ps[n // 2] @ geometry.rigid_transform(tx=0.02 * k, rz=1e-4 * k)).T.reshape(12)
        poses.append(Pk)
    m = E.MetricRadonIntermediate(ctx, Ps, dtrs).setSampling("polynomial")
    ref = E.MetricRadonIntermediate(ctx, Ps, dtrs).setSampling("polynomial").setSmallEval(False)
    n_pairs = n * (n - 1) // 2
    L = min(n_pairs, 150)
    ab = [(i, j) for i in range(n) for j in range(i + 1, n)][:L]
    idx = np.array([(a, b, a, b) for a, b in ab], np.int32)
    out, out_ref = np.empty(L, np.float32), np.empty(L, np.float32)
    want_all = [ref.setProjectionMatrices(Pk).evaluate() for Pk in poses]
    want_list = []
    for Pk in poses:
        want_list.append((ref.setProjectionMatrices(Pk).evaluate(idx, out_ref), out_ref.copy()))
    share = total // 5
    for it in range(share):
        k = it & 31
        m.setProjectionMatrices(poses[k])
        if it & 1:
            v = m.evaluate()
            ok = v == want_all[k]
        else:
            v = m.evaluate(idx, out)
            ok = v == want_list[k][0] and np.array_equal(out, want_list[k][1])
        if not ok:
            bad += 1
            if bad < 10:
                print("MISMATCH n=%d it=%d kind=%d got %r" % (n, it, it & 1, v), flush=True)
        done += 1
        if done % 200000 == 0:
            print("%d evaluations, %d mismatches, %.1f s" % (done, bad, time.time() - t0), flush=True)
    m.close(); ref.close()
print("soak: %d evaluations, %d mismatches, %.1f s" % (done, bad, time.time() - t0))
sys.exit(1 if bad else 0)
