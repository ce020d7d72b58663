"""Experiment: the XCDs' shares of an all-pairs launch.  pairs_kernel gives XCD x the x-th contiguous eighth of the pair order
(wave w of local block b: position w * nblk + x * per_xcd + b); the kappa_max = pi/2 pairs sit at the ends of the first rows
of the pair triangle, so XCD 0 gets 547 of them and XCD 7 182 (scripts/experiments/exp_wave_timeline.py: XCD 0 finishes 9 %
after XCD 7).  Here the same kernel runs over index lists whose ORDER emulates other assignments:
  natural    the get_ij order (what the all-pairs launch does)
  snake      whole rows of the pair triangle dealt to the XCDs in boustrophedon order (rows 0..7 -> XCD 0..7, rows 8..15 -> XCD 7..0,
             ...): every XCD gets the same number of pairs and of heavy pairs, consecutive pairs of a row stay on one XCD
  snake_q    the same, and the four waves of a workgroup take their XCD's list a quarter apart (as the product does in its eighth)
Kernel time by HIP events.  python scripts/exp_xcd_balance.py"""
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
m = E.MetricRadonIntermediate(ctx, Ps, dtrs).setSampling("polynomial")
nblk = (N + 3) // 4
per_xcd = (nblk + 7) // 8
row_start = np.concatenate([[0], np.cumsum(n - 1 - np.arange(n - 1))])  # first pair of row i


def xcd_lists(group):
    """rows dealt to the 8 XCDs in snake order, `group` rows at a time; then lengths evened out by moving tail pairs"""
    lists = [[] for _ in range(8)]
    k = 0
    for r0 in range(0, n - 1, group):
        x = k % 16
        x = x if x < 8 else 15 - x
        for r in range(r0, min(r0 + group, n - 1)):
            lists[x].append(np.arange(row_start[r], row_start[r + 1]))
        k += 1
    lists = [np.concatenate(l) for l in lists]
    # even out: every list at most 4 * per_xcd long and the total preserved
    cap = [4 * min(per_xcd, nblk - x * per_xcd) for x in range(8)]  # blocks XCD x really has
    spill = []
    for x in range(8):
        if len(lists[x]) > cap[x]:
            spill.append(lists[x][cap[x]:]); lists[x] = lists[x][:cap[x]]
    spill = np.concatenate(spill) if spill else np.empty(0, np.int64)
    for x in range(8):
        room = cap[x] - len(lists[x])
        if room > 0 and len(spill):
            lists[x] = np.concatenate([lists[x], spill[:room]]); spill = spill[room:]
    assert len(spill) == 0
    return lists


def emulate(lists, quarters):
    """list order such that the kernel's mapping gives XCD x the pairs lists[x]: position w * nblk + x * per_xcd + b"""
    order = np.full(4 * nblk, -1, np.int64)
    for x in range(8):
        L = lists[x]
        q = (len(L) + 3) // 4
        assert q <= min(per_xcd, nblk - x * per_xcd)
        for w in range(4):
            seg = L[w * q:(w + 1) * q] if quarters else L[w::4]
            lo = w * nblk + x * per_xcd
            order[lo:lo + len(seg)] = seg
    return order


def run(order):
    live = order[order >= 0]
    assert len(np.unique(live)) == N
    # holes (positions without a pair) cannot be expressed in a list: fill them with a light pair of the last row
    o = order[:N].copy()
    missing = np.setdiff1d(np.arange(N), o[o >= 0])
    o[o < 0] = missing[:int((o < 0).sum())] if len(missing) else N - 1
    idx4 = np.stack([iu[0][o], iu[1][o], iu[0][o], iu[1][o]], 1).astype(np.int32)
    vals = np.empty(N, np.float32)
    ctx.enable_timing(True)
    ks = []
    for _ in range(17):
        m.evaluate(idx4, vals)
        ks.append(ctx.last_kernel_ms("pairs"))
    ctx.enable_timing(False)
    return 1e3 * float(np.median(ks[3:])), len(np.unique(o))


deg = np.concatenate([[q["degree"] for q in m.debug_polynomials(a, min(10000, N - a))] for a in range(0, N, 10000)])
heavy = deg == 0


def class_aware(lists, mode):
    """within every XCD's list: heavy pairs first ("first"), or at evenly spaced positions ("spread"), or evenly over the first
    85 % of the list ("spread85": none in the launch's tail); everything else keeps its order"""
    out_l = []
    for L in lists:
        h = L[heavy[L]]; l = L[~heavy[L]]
        if mode == "first":
            out_l.append(np.concatenate([h, l])); continue
        span = len(L) if mode == "spread" else int(0.85 * len(L))
        # positions in the QUARTER layout: wave w takes L[w*q:(w+1)*q], so "time" of list index t is t % q; spread over time AND waves
        q = (len(L) + 3) // 4
        tpos = (np.arange(len(h)) * (span / max(len(h), 1))).astype(np.int64)  # index in a time-major enumeration
        # time-major enumeration e -> list index: block b = e // 4, wave w = e % 4 -> w*q + b
        li = (tpos % 4) * q + (tpos // 4)
        li = np.minimum(li, len(L) - 1)
        li = np.unique(li)
        mask = np.zeros(len(L), bool); mask[li[:len(h)]] = True
        extra = len(h) - int(mask.sum())
        if extra > 0:
            free = np.flatnonzero(~mask)[:extra]; mask[free] = True
        o = np.empty(len(L), np.int64); o[mask] = h; o[~mask] = l
        out_l.append(o)
    return out_l


out = {}
for rep in range(3):
    for name, order in (("natural", np.arange(4 * nblk)[: 4 * nblk] * 1),
                        ("snake1_q", emulate(xcd_lists(1), True)), ("snake2_q", emulate(xcd_lists(2), True)),
                        ("snake1_strided", emulate(xcd_lists(1), False)),
                        ("snake1_q_heavy_first", emulate(class_aware(xcd_lists(1), "first"), True)),
                        ("snake1_q_heavy_spread", emulate(class_aware(xcd_lists(1), "spread"), True)),
                        ("snake1_q_heavy_spread85", emulate(class_aware(xcd_lists(1), "spread85"), True))):
        if name == "natural":
            order = np.where(np.arange(4 * nblk) < N, np.arange(4 * nblk), -1)
        us, distinct = run(order)
        out.setdefault(name, []).append(dict(kernel_us=us, distinct_pairs=distinct))
print(json.dumps(out, indent=1))
