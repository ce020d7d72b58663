"""Bank-conflict simulation of the Radon kernel's CURRENT LDS layout (round 3 on): element (i, r) = texel pair {T(i, r), T(i+1, r)}
(8 bytes) at pair index (r - R0) * S + (i - I0), fetched with ds_read_b64 -- serviced in the two 32-lane halves of a wave, 32
pair slots (64 banks x 4 B) per LDS cycle, identical addresses broadcast, cycles of a half = the largest number of DISTINCT
addresses on one slot (MI355X_MICROARCH.md, LDS).  Question: is there a bank function or a lane
arrangement that costs no vector instruction and separates what a half reads?  Variables: the row stride S modulo 32 (a
row-dependent slot shift is free: S is a per-slab constant already) and which (angle, distance) bins share a half.
    python scripts/analysis/radon_lds_b64_sim.py
Prints the mean LDS cycles per half-wave read (1.0 = conflict-free; the kernel's measured SQ_LDS_BANK_CONFLICT /
SQ_LDS_IDX_ACTIVE = 0.47 corresponds to 1.89)."""
import numpy as np
rng = np.random.default_rng(0)
n_alpha = n_t = 768
W = H = 1024
D = np.sqrt(2) * 1024


def half_cycles(addr):
    slot = addr % 32
    c = 1
    for s in np.unique(slot):
        c = max(c, len(np.unique(addr[slot == s])))
    return c


def sim(arr, q, thetas, trials=24, steps=6):
    """arr(ia0, it0, w) -> (angle index, distance index) of the 64 lanes of wave w; q = S mod 32."""
    tot = n = 0
    for th in thetas:
        for _ in range(trials):
            ia0 = int((th / np.pi + 0.5) * n_alpha) // 16 * 16
            it0 = int(rng.integers(4, n_t // 32 - 4)) * 32
            a_idx, t_idx = arr(ia0, it0, int(rng.integers(0, 4)))
            alpha = (a_idx / n_alpha - 0.5) * np.pi
            tau = (t_idx / n_t - 0.5) * D
            l0, l1 = -np.sin(alpha), np.cos(alpha)
            l2 = -tau - 0.5 * W * l0 - 0.5 * H * l1
            ox, oy, dx, dy = -l2 * l0, -l2 * l1, l1, -l0
            t = rng.uniform(-350, 350) + rng.uniform(0, 0.66, 64)  # the lanes' own t grids are offset by their clip points
            transp = abs(l1.mean()) > abs(l0.mean())  # fast axis = the image axis closer to the line normal
            S = 96 + q
            for i in range(steps):
                for s in (0.5, -0.5):  # the two samples of the derivative pair: +- half a pixel along the normal
                    x = ox + (t + 0.66 * i) * dx + 0.5 + s * l0
                    y = oy + (t + 0.66 * i) * dy + 0.5 + s * l1
                    fi = np.floor(x - 0.5).astype(np.int64) + 4096
                    fj = np.floor(y - 0.5).astype(np.int64) + 4096
                    f, r = (fj, fi) if transp else (fi, fj)
                    for rr in (0, 1):  # the footprint's two rows: two ds_read_b64
                        addr = (r + rr) * S + f
                        for h in (0, 32):
                            tot += half_cycles(addr[h:h + 32])
                            n += 1
    return tot / n


arrs = {
    "2 adjacent angles x 16 distances per half (current)": lambda ia0, it0, w: (ia0 + 4 * w + np.repeat(np.arange(4), 16), it0 + np.tile(np.arange(16), 4)),
    "1 angle x 32 distances per half": lambda ia0, it0, w: (ia0 + 2 * w + np.repeat(np.arange(2), 32), it0 + np.tile(np.arange(32), 2)),
    "angles a, a+8 x 16 distances per half": lambda ia0, it0, w: (ia0 + np.repeat(np.array([2 * w, 2 * w + 8, 2 * w + 1, 2 * w + 9]), 16), it0 + np.tile(np.arange(16), 4)),
    "4 angles x 8 distances per half": lambda ia0, it0, w: (ia0 + 8 * (w % 2) + np.repeat(np.arange(8), 8), it0 + 8 * (w // 2) + np.tile(np.arange(8), 8)),
    "1 angle x 16 even + 16 odd distances of 32": lambda ia0, it0, w: (ia0 + 2 * w + np.repeat(np.arange(2), 32), it0 + np.tile(np.concatenate([np.arange(0, 32, 2), np.arange(1, 32, 2)]), 2)),
}
thetas = np.linspace(-np.pi / 2 + 0.02, np.pi / 2 - 0.02, 10)
for name, arr in arrs.items():
    res = {q: sim(arr, q, thetas) for q in (0, 1, 2, 4, 8, 12, 16, 17, 24)}
    best = min(res, key=res.get)
    print("%-52s S mod 32 = 0: %.2f cycles per half;  best S mod 32 = %2d: %.2f;  all: %s"
          % (name, res[0], best, res[best], " ".join("%d:%.2f" % (q, v) for q, v in res.items())))
