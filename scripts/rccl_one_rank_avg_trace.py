#!/usr/bin/env python3
"""One-rank RCCL group, the sharded evaluation with the collective's reduction op set to AVG instead of SUM (GPU box; run under
rocprofv3 --kernel-trace by scripts/rccl_first_contact.sh).  With ONE rank RCCL completes an in-place SUM all-reduce without
launching anything; for AVG (a pre-multiplied sum: x * 1/1) it launches its one-rank reduction kernel -- the only way to see RCCL
DEVICE code run in the context's stream between pairs_kernel / sum_pairs_kernel and publish_scalar_kernel on a one-GPU box.
The value must be unchanged (x * 1.0).  Not the product's op: sharding.allreduce_mean uses SUM."""
import os, socket, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic

with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
torch.cuda.set_stream(torch.cuda.Stream(dev))
ctx = E.Context(0, stream=torch.cuda.current_stream().cuda_stream)
n, S, B = 150, 256, 192
Ps = synthetic.short_scan(n, S, S, 0.308 * 1024.0 / S)
imgs = synthetic.projections_torch(Ps, S, S, synthetic.sphere_phantom(), dev)
slabs = torch.zeros((n, E.slab_floats(B, B)), dtype=torch.float32, device=dev)
keep = E.RadonIntermediate.compute_into(ctx, imgs, slabs, B, B)
ctx.synchronize()
dtrs = [E.RadonIntermediate.wrap_device(ctx, slabs[k], B, B, S, S) for k in range(n)]
m = E.MetricRadonIntermediate(ctx, Ps, dtrs)
n_pairs = n * (n - 1) // 2
want = m.evaluate()
t = torch.zeros(1, dtype=torch.float64, device=dev)
got = []
for rep in range(5):
    m.setProjectionMatrices(Ps)
    m.evaluate_range_async(0, n_pairs, t)
    dist.all_reduce(t, op=dist.ReduceOp.AVG)
    m.publish_scalar(t)
    got.append(m.wait_scalar() / n_pairs)
print("one-rank RCCL all-reduce(AVG) between the sum and the publish kernel: %d pairs, mean %.17g, plain evaluate %.17g, equal %s"
      % (n_pairs, got[-1], want, all(g == want for g in got)))
dist.barrier()
dist.destroy_process_group()
