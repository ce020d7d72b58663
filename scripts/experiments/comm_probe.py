import os, sys, time, socket
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.distributed as dist
import epipolarconsistency_amd as E
from epipolarconsistency_amd import sharding, synthetic, _lib
import ctypes as C
n, S, B = 400, 1024, 768
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
mode = sys.argv[1]
if mode != "nodist":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    t = torch.ones(1, device=dev); dist.all_reduce(t)
Ps = synthetic.short_scan(n, S, S, 0.308); ph = synthetic.sphere_phantom()
torch.cuda.set_stream(torch.cuda.Stream(dev))
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
for a in range(0, n, 50):
    imgs = synthetic.projections_torch(Ps[a:a + 50], S, S, ph, dev)
    keep = E.RadonIntermediate.compute_into(ctx, imgs, slabs[a:a + 50], B, B); ctx.synchronize()
dtrs = [E.RadonIntermediate.wrap_device(ctx, slabs[k], B, B, S, S) for k in range(n)]
m = E.MetricRadonIntermediate(ctx, Ps, dtrs)
P = E.pack_projection_matrices(Ps); P2 = P.copy(); P2.reshape(-1)[200 * 12 + 9] += 1e-3
N = n * (n - 1) // 2
def run(tag, f):
    for k in range(30): m.setProjectionMatrices(P2 if k & 1 else P); f()
    ctx.enable_timing(True); ks = []
    for k in range(30): m.setProjectionMatrices(P2 if k & 1 else P); f(); ks.append(ctx.last_kernel_ms("pairs"))
    ctx.enable_timing(False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(600): m.setProjectionMatrices(P2 if k & 1 else P); f()
    torch.cuda.synchronize(); dt = 1e6 * (time.perf_counter() - t0) / 600
    st = np.zeros(8); _lib.lib().ecc_debug_step_stamps(m._h, C.c_void_p(st.ctypes.data))
    print(tag, "step %.1f us, kernel %.1f us, stamps(us) %s" % (dt, 1e3 * np.median(ks), np.round(1e6 * (st[1:] - st[0]), 1)))
if mode == "commfirst":
    comm = sharding.RcclComm(ctx, 0, 1, sharding.torch_broadcast_bytes(dev))
    run("comm created first: evaluate_range", lambda: m.evaluate_range(0, N))
    print("empty allreduce", m.evaluate_range_allreduce(comm, 0, 0))
    run("after the empty all-reduce: evaluate_range", lambda: m.evaluate_range(0, N))
    run("allreduce", lambda: m.evaluate_range_allreduce(comm, 0, N))
    sys.exit(0)
run("plain evaluate_range", lambda: m.evaluate_range(0, N))
if mode == "comm":
    comm = sharding.RcclComm(ctx, 0, 1, sharding.torch_broadcast_bytes(dev))
    run("with comm alive: evaluate_range", lambda: m.evaluate_range(0, N))
    run("with comm alive: allreduce", lambda: m.evaluate_range_allreduce(comm, 0, N))
    comm.close()
    run("comm destroyed: evaluate_range", lambda: m.evaluate_range(0, N))
