"""Experiment: the ORDER in which the pair kernel takes the 79 800 pairs of the BASELINE workload, measured the way the
benchmark runs -- SUSTAINED, i.e. at the socket's power cap (a variant runs back to back for `secs` seconds; kernel time by HIP
events and socket power / engine clock from hwmon over the last 60 % of them).  Short runs (a few launches per variant, as
scripts/exp_xcd_balance.py does) see the chip at its top clock, where the XCDs' finish times matter; at the cap they do not.
The orders are emulated with index lists over the same kernel (position w * nblk + x * per_xcd + b -> wave w of local block b
of XCD x):
  natural        get_ij order
  snake          rows of the pair triangle dealt to the XCDs in boustrophedon order (equal shares of every cost class)
  tileT          T x T-view tiles of the pair triangle, tile by tile in get_ij order of the tiles (halves the L2 misses, round 1)
  tileT_snake    the tiles dealt to the XCDs in boustrophedon order
python scripts/exp_orders_sustained.py [secs]"""
import glob, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import epipolarconsistency_amd as E
from epipolarconsistency_amd import synthetic
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
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
m = E.MetricRadonIntermediate(ctx, Ps, dtrs).setSampling("polynomial")
N = n * (n - 1) // 2
iu = np.triu_indices(n, 1)
nblk = (N + 3) // 4
per_xcd = (nblk + 7) // 8
blocks_x = [min(per_xcd, nblk - x * per_xcd) for x in range(8)]


def deal(units):
    """units (arrays of pair indices) dealt to 8 lists in boustrophedon order, lengths evened out to the XCDs' block counts"""
    lists = [[] for _ in range(8)]
    for k, u in enumerate(units):
        x = k % 16
        lists[x if x < 8 else 15 - x].append(u)
    lists = [np.concatenate(l) if l else np.empty(0, np.int64) for l in lists]
    spill = []
    for x in range(8):
        cap = 4 * blocks_x[x]
        if len(lists[x]) > cap:
            spill.append(lists[x][cap:]); lists[x] = lists[x][:cap]
    spill = np.concatenate(spill) if spill else np.empty(0, np.int64)
    for x in range(8):
        room = 4 * blocks_x[x] - len(lists[x])
        if room > 0 and len(spill):
            lists[x] = np.concatenate([lists[x], spill[:room]]); spill = spill[room:]
    assert len(spill) == 0
    return lists


def emulate(lists):
    order = np.full(4 * nblk, -1, np.int64)
    for x in range(8):
        L = lists[x]
        q = (len(L) + 3) // 4
        for w in range(4):
            seg = L[w * q:(w + 1) * q]
            order[w * nblk + x * per_xcd:w * nblk + x * per_xcd + len(seg)] = seg
    return order


def contiguous(seq):
    """a plain list order: the kernel's own mapping (XCD x = the x-th eighth of the positions of each quarter)"""
    order = np.full(4 * nblk, -1, np.int64)
    order[:N] = seq
    return order


row_start = np.concatenate([[0], np.cumsum(n - 1 - np.arange(n - 1))])
rows = [np.arange(row_start[r], row_start[r + 1]) for r in range(n - 1)]
pid = np.full((n, n), -1, np.int64)
pid[iu] = np.arange(N)


def tiles(T):
    out = []
    for a in range(0, n, T):
        for b in range(a, n, T):
            blk = pid[a:a + T, b:b + T]
            v = blk[blk >= 0]
            if len(v):
                out.append(v)
    return out


def interleave(R):
    """R consecutive rows of the pair triangle walked together: (i, j), (i + 1, j), ..., (i + R - 1, j), then j + 1: a view j's band
    is used by R pairs in a row (its next use is otherwise a whole row of pairs later, long after the L2 has dropped it)"""
    out = []
    for a in range(0, n - 1, R):
        block = pid[a:a + R, :]
        cols = block.T.reshape(-1)  # j-major: for every j the R rows
        out.append(cols[cols >= 0])
    return np.concatenate(out)


orders = {"natural": contiguous(np.arange(N)), "snake": emulate(deal(rows)),
          "rows2": contiguous(interleave(2)), "rows4": contiguous(interleave(4)), "rows8": contiguous(interleave(8)),
          "tile16": contiguous(np.concatenate(tiles(16))), "tile32": contiguous(np.concatenate(tiles(32))),
          "tile16_snake": emulate(deal(tiles(16))), "tile32_snake": emulate(deal(tiles(32)))}

hw = None
try:
    pr = torch.cuda.get_device_properties(dev)
    pci = "%04x:%02x" % (pr.pci_domain_id, pr.pci_bus_id)
except Exception:
    pci = None
cands = [d for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")) if os.path.exists(d + "/power1_input") and os.path.exists(d + "/freq1_input")]
for d in cands:
    if pci and pci in os.path.realpath(os.path.dirname(os.path.dirname(d))):
        hw = d
if hw is None and len(cands) == 1:
    hw = cands[0]
sys.stderr.write("hwmon candidates %r, device %r -> %r\n" % (cands, pci, hw))
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
if only:
    orders = {k: v for k, v in orders.items() if k in only}


def run(order):
    o = order[:N].copy()
    assert (o >= 0).all() and len(np.unique(o)) == N
    idx4 = np.stack([iu[0][o], iu[1][o], iu[0][o], iu[1][o]], 1).astype(np.int32)
    vals = np.empty(N, np.float32)
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
    ks = []
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < secs:
        mean = m.evaluate(idx4, vals)
        ks.append((time.perf_counter() - t0, ctx.last_kernel_ms("pairs")))
    ctx.enable_timing(False)
    stop[0] = True
    if hw:
        th.join()
    late = [k for t, k in ks if t > 0.4 * secs]
    ps = [(p, f) for t, p, f in samples if t - t0 > 0.4 * secs]
    return dict(kernel_us=1e3 * float(np.median(late)), launches=len(ks), mean=mean,
                watts=float(np.mean([p for p, f in ps])) if ps else None, sclk_mhz=float(np.mean([f for p, f in ps])) if ps else None)


out = {}
for rep in range(2):
    for name, order in orders.items():
        r = run(order)
        out.setdefault(name, []).append(r)
        sys.stderr.write("%s %r\n" % (name, r))
print(json.dumps(out, indent=1))
