// ecc_slab_tile.h -- line integrals through an LDS tile of texel pairs: what the Radon kernel (radon_kernel.hip) and
// MetricDirect's line-integral kernel (direct_kernel.hip) share.
//
// A workgroup of 256 threads owns 256 nearly parallel lines; each thread steps along its own line and accumulates
// bilinear samples sequentially (the reference's order).  The lines sweep a narrow band of the image, which is cut into
// SLABS across the image axis the lines run along (the "slow" axis s; the other one is the "fast" axis f):
//   * a slab lives in LDS as TEXEL PAIRS, element (i, r) = {T(i, r), T(i+1, r)} (clamp addressing resolved while
//     staging): one aligned ds_read_b64 fetches a footprint row (256 B/clk/CU);
//   * the tile is a LINEAR array, address(i, r) = (r - R0) * S + (i - I0), the row stride S a multiple of 32 pairs
//     (bank = i mod 32 whatever the row).  Row r only ever holds the S - 1 columns from ws(r) = floor(alpha + beta * r)
//     on, beta = the band's slope df/ds: the window slides with the band, a slab stores a parallelogram, not the band's
//     bounding box, and S only has to cover the band's width ALONG f (64 ... 256, per slab).  A slab is
//     floor(TILE_CAP / (S + 1)) rows thick whatever the angle;
//   * slab geometry comes from a band description that is identical in every thread (no reductions per slab, no
//     decisions through LDS).  It does not have to be trusted: every thread checks the two end points of its own run
//     through the slab against the slab's admissible region (two half-planes in the sheared coordinate f - beta*s, two
//     in s: a convex set, and positions are linear in t) and samples from global memory when the check fails;
//   * the next slab's texels are loaded into registers BEFORE the current slab is sampled and stored to LDS after it:
//     two barriers per slab and no exposed global-memory latency;
//   * float -> LDS address is one fp32 add of 2^23 and a shift-add on the bits.
// Arithmetic of a sample: the exact fp32 bilinear rule of ecc_sampling.h, bit for bit.
#ifndef ECC_SLAB_TILE_H
#define ECC_SLAB_TILE_H

#include <hip/hip_runtime.h>
#include <float.h>

#include "ecc_sampling.h"

#ifndef ECC_SLAB_STAT
#define ECC_SLAB_STAT(i, v)
#endif

namespace ecc_slab {

constexpr int THREADS = 256;               // threads (= lines) per workgroup
constexpr int WAVES = THREADS / 64;
#ifndef ECC_SLAB_TILE_CAP
#define ECC_SLAB_TILE_CAP 5056
#endif
#ifndef ECC_SLAB_N_PRE
#define ECC_SLAB_N_PRE 26
#endif
constexpr int TILE_CAP = ECC_SLAB_TILE_CAP; // texel pairs per workgroup: 40 448 B, four workgroups per CU
constexpr int N_PRE = ECC_SLAB_N_PRE;       // texels a thread stages per slab (registers that live across the sampling loop)
static_assert(N_PRE <= 32, "stage_regs is a 32-float vector");
constexpr int S_MIN = 64, S_MAX = 256;     // row stride of the tile in pairs (multiples of 32)
constexpr int MAX_SLABS = 8192;            // bound on the slab loop (every loop is bounded)
constexpr float MAGIC = 8388608.f;         // 2^23: as_uint(k + 2^23) = 0x4B000000 + k for integers 0 <= k < 2^23

typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2f lds_v2f;
typedef float stage_regs __attribute__((ext_vector_type(32)));  // a vector value, not an array: never addressed, so never in scratch

__device__ __forceinline__ float wave_min_f(float v)
{
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v)
{
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

struct Shared {
    float2 tile[TILE_CAP];
    float red[WAVES][4];  // workgroup-wide minima / maxima at start-up
};

// values every lane computes identically, moved to scalar registers (branches on them become scalar branches)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// What every band description has in common (all values identical in every thread).
struct BandFrame {
    float beta, abs_beta;  // slope df/ds the row windows follow (clamped to [-1, 1])
    int bstep;             // beta in 1/65536 columns per row: the row windows are placed by integer arithmetic
    float sigma;           // +1: s grows with t, -1: s falls
    float marg;            // allowance on either side of the band (lines in between, rounding)
};
__device__ __forceinline__ void set_beta(BandFrame& f, float beta)
{
    f.beta = uni(fminf(fmaxf(beta, -1.f), 1.f));
    f.abs_beta = fabsf(f.beta);
    f.bstep = uni((int)rintf(f.beta * 65536.f));
}

struct SlabPlan {
    int R0, H, S, I0;      // first staged row, rows, row stride (pairs), column of address 0
    int acc0;              // row window of tile row rr: ws = (acc0 + rr * bstep) >> 16 = floor(alpha + beta * (R0 + rr)) up to
                           // 1e-3 columns, S - 1 elements from there
    float Glo, Ghi;        // admissible f - beta * s of a sample
    float slo, shi;        // admissible s of a sample
    float b_next;          // samples with s up to b_next (in walking order) belong to this slab
};

// Rows of a slab by k = S / 32: the linear tile needs H * (S + 1) pairs (the row windows slide by up to one column per
// row); wave w stages rows w, w + 4, ..., a row in ceil(S / 64) loads of 64 consecutive texels, N_PRE registers in all.
constexpr int slab_rows_of(int k)
{
    return (TILE_CAP / (32 * k + 1)) < WAVES * (N_PRE / ((k + 1) / 2)) ? (TILE_CAP / (32 * k + 1)) : WAVES * (N_PRE / ((k + 1) / 2));
}
constexpr unsigned long long SLAB_ROWS_PACKED = (unsigned long long)slab_rows_of(2) | ((unsigned long long)slab_rows_of(3) << 8) |
                                                ((unsigned long long)slab_rows_of(4) << 16) | ((unsigned long long)slab_rows_of(5) << 24) |
                                                ((unsigned long long)slab_rows_of(6) << 32) | ((unsigned long long)slab_rows_of(7) << 40) |
                                                ((unsigned long long)slab_rows_of(8) << 48);
__device__ __forceinline__ int slab_rows(int k) { return (int)((SLAB_ROWS_PACKED >> (8 * (k - 2))) & 255ull); }
static_assert(S_MIN == 64 && S_MAX == 256 && slab_rows_of(2) < 256, "slab_rows covers k = 2 .. 8");


constexpr int SLAB_SPAN = slab_rows_of(S_MIN / 32) - 5;  // the thickest slab: what a band's extent is evaluated over

// Slab that starts at slow coordinate b (walking order).  band(sa, sb, gmin, gmax): extent of the band's sheared
// coordinate f - beta * s over slow coordinates [sa, sb].  Every thread computes the same plan.
template <class Band>
__device__ __forceinline__ SlabPlan plan_slab(const BandFrame& bf, const Band& band, float b)
{
    SlabPlan sp;
    const float span = (float)SLAB_SPAN;
    const float sa = (bf.sigma > 0.f ? b : b - span) - 1.f, sb = sa + span + 2.f;
    float gmin, gmax;
    band(sa, sb, gmin, gmax);
    const float need = (gmax - gmin) + 2.f * bf.marg + 2.f + 3.f * bf.abs_beta + .25f;
    const int S = uni(min(max(((int)ceilf(fminf(need, 4096.f)) + 31) & ~31, S_MIN), S_MAX));
    const int H = slab_rows(S >> 5);
    const float delta = (float)(H - 5);
    const float slack = (float)S - need;  // negative: the band does not fit, the per-thread checks decide
    sp.S = S;
    sp.H = H;
    const float alpha = gmin - bf.marg - .5f - 1.5f * bf.abs_beta - .125f - .5f * slack;
    sp.Glo = alpha + .5f + 1.5f * bf.abs_beta + .03f;
    sp.Ghi = alpha + (float)S - 1.5f - 1.5f * bf.abs_beta - .03f;
    const float lo = bf.sigma > 0.f ? b : b - delta;
    sp.R0 = uni((int)floorf(lo - 1.75f));
    sp.slo = (float)sp.R0 + .51f;
    sp.shi = (float)(sp.R0 + H) - .51f;
    sp.b_next = uni(b + bf.sigma * delta);
    sp.acc0 = uni((int)floorf(__builtin_fmaf(bf.beta, (float)sp.R0, alpha) * 65536.f));
    sp.I0 = min(sp.acc0 >> 16, (sp.acc0 + (H - 1) * bf.bstep) >> 16);
    return sp;
}

// rows a wave stages when a row takes K loads: ceil(rows of the smallest such stride / 4)
constexpr int stage_rows_of(int K) { return (slab_rows_of(K == 1 ? 2 : 2 * K - 1) + WAVES - 1) / WAVES; }
static_assert(stage_rows_of(1) * 1 <= N_PRE && stage_rows_of(2) * 2 <= N_PRE && stage_rows_of(3) * 3 <= N_PRE && stage_rows_of(4) * 4 <= N_PRE, "N_PRE");

template <int K>
__device__ __forceinline__ void stage_load_k(const SlabPlan& sp, int bstep, const float* __restrict__ src, int Nf, int Ns,
                                             stage_regs& reg)
{
    const int w = uni((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    int acc = sp.acc0 + w * bstep, r = sp.R0 + w;  // uniform
#pragma unroll
    for (int j = 0; j < stage_rows_of(K); ++j) {
        // rows past the slab's last one (and columns past a row's window) are loaded and never stored
        const int rowoff = min(max(r, 0), Ns - 1) * Nf;
        const int col0 = (acc >> 16) + lane;
#pragma unroll
        for (int seg = 0; seg < K; ++seg) {
            const int ic = min(max(col0 + seg * 64, 0), Nf - 1);
            reg[j * K + seg] = src[(unsigned)(rowoff + ic)];
        }
        acc += WAVES * bstep;
        r += WAVES;
    }
}

// Staging, second half: texel c of a row is the .x of pair c and the .y of pair c - 1; texel S - 1 has no pair of its
// own (the next row's window may start one column earlier), texel 0 no left neighbour.
template <int K>
__device__ __forceinline__ void stage_store_k(const SlabPlan& sp, int bstep, unsigned tile_addr, const stage_regs& reg)
{
    typedef __attribute__((address_space(3))) float lds_float;
    const int w = uni((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    int acc = sp.acc0 + w * bstep, rr = w;  // uniform
    const unsigned lane8 = 8u * (unsigned)lane;
    const int last = sp.S - 64 * (K - 1);  // columns of the row's last load
#pragma unroll
    for (int j = 0; j < stage_rows_of(K); ++j) {
        if (rr < sp.H) {
            // byte address of the .y of pair (c - 1) for this lane's c of load 0
            const unsigned a = tile_addr + 8u * (unsigned)(rr * sp.S + ((acc >> 16) - sp.I0)) - 4u + lane8;
#pragma unroll
            for (int seg = 0; seg < K; ++seg) {
                const float v = reg[j * K + seg];
                const bool px = seg < K - 1 || lane < last - 1;  // c < S - 1
                const bool py = (seg > 0 || lane > 0) && (seg < K - 1 || lane < last);  // 0 < c < S
                if (px) *(lds_float*)(size_t)(a + 512u * seg + 4u) = v;
                if (py) *(lds_float*)(size_t)(a + 512u * seg) = v;
            }
        }
        acc += WAVES * bstep;
        rr += WAVES;
    }
}

__device__ __forceinline__ void stage_load(const SlabPlan& sp, int bstep, const float* __restrict__ src, int Nf, int Ns,
                                           stage_regs& reg)
{
    switch ((sp.S + 63) >> 6) {  // uniform
    case 1: stage_load_k<1>(sp, bstep, src, Nf, Ns, reg); break;
    case 2: stage_load_k<2>(sp, bstep, src, Nf, Ns, reg); break;
    case 3: stage_load_k<3>(sp, bstep, src, Nf, Ns, reg); break;
    default: stage_load_k<4>(sp, bstep, src, Nf, Ns, reg); break;
    }
}
__device__ __forceinline__ void stage_store(const SlabPlan& sp, int bstep, unsigned tile_addr, const stage_regs& reg)
{
    switch ((sp.S + 63) >> 6) {
    case 1: stage_store_k<1>(sp, bstep, tile_addr, reg); break;
    case 2: stage_store_k<2>(sp, bstep, tile_addr, reg); break;
    case 3: stage_store_k<3>(sp, bstep, tile_addr, reg); break;
    default: stage_store_k<4>(sp, bstep, tile_addr, reg); break;
    }
}
static_assert(S_MAX <= 256, "stage_load / stage_store dispatch on ceil(S / 64) = 1 .. 4");

// The exact bilinear rule (ecc_sampling.h) on the pair tile.  base = LDS byte address of the tile, minus
// 8 * (R0 * S + I0) (tile origin), minus (0x4B000000 << 3) (the 2^23 trick), all modulo 2^32.
// TRANSP: f is the image's y axis: pair (j, i) = {T(i, j), T(i, j+1)}, the next row is i + 1.
template <bool TRANSP, bool FMA>
__device__ __forceinline__ float tex_pairs(unsigned base, float Sf, unsigned S8, float x, float y)
{
    float xb = x - 0.5f, yb = y - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    float fx = xb - fi, fy = yb - fj;
    // row * S + column: an exact small non-negative integer in fp32 whether or not it is fused (an explicit fma: the
    // build runs with -ffp-contract=off); adding 2^23 leaves it in the low mantissa bits
    const float idx = TRANSP ? __builtin_fmaf(fi, Sf, fj) : __builtin_fmaf(fj, Sf, fi);
    const unsigned a0 = (__float_as_uint(idx + MAGIC) << 3) + base;
    const unsigned a1 = a0 + S8;  // a run-time stride: two ds_read_b64 (256 B/clk each), never one ds_read2_b64 (128 B/clk)
    // (Measured and dropped: loops specialised on the stride -- 64 / 96 / 128 -- with the second read as inline assembly
    // at an immediate offset and one explicit s_waitcnt: 50 instead of 52 vector instructions per step, bit-exact, and
    // 0.580 against 0.573 ms per image: the full drain before the first product costs more than the add.)
    const v2f pa = *(const lds_v2f*)(size_t)a0;
    const v2f pb = *(const lds_v2f*)(size_t)a1;
    const float T00 = pa.x, T10 = TRANSP ? pb.x : pa.y;
    const float T01 = TRANSP ? pa.y : pb.x, T11 = pb.y;
    return ecc_bilerp<FMA>(fx, fy, T00, T10, T01, T11);
}

// position(t) of a line: o + t * d, rounded twice (exact convention) or once (contracted)
template <bool FMA>
__device__ __forceinline__ float line_pos(float o, float t, float d)
{
    return FMA ? __builtin_fmaf(t, d, o) : o + t * d;
}


// One thread's line: position(t) = (o0, o1) + t * (d0, d1) in image coordinates (texel centres at i + 0.5), samples at
// t, t + step, ... while t <= t_max.
struct LineRun {
    float o0, o1, d0, d1, t, t_max;
    bool active;
};

// Workgroup-wide minimum of lo and maximum of hi (threads that do not take part pass FLT_MAX / -FLT_MAX); one barrier.
__device__ __forceinline__ void block_min_max(Shared& sh, float& lo, float& hi)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lo = wave_min_f(lo);
    hi = wave_max_f(hi);
    if (lane == 0) {
        sh.red[wave][0] = lo;
        sh.red[wave][1] = hi;
    }
    __syncthreads();
    lo = sh.red[0][0];
    hi = sh.red[0][1];
#pragma unroll
    for (int q = 1; q < WAVES; ++q) {
        lo = fminf(lo, sh.red[q][0]);
        hi = fmaxf(hi, sh.red[q][1]);
    }
    lo = uni(lo);
    hi = uni(hi);
    __syncthreads();  // sh.red may be reused
}

// Walks every thread's line through the slabs of the workgroup's band.  Per step the samples A = position + (a0, a1)
// (A_OFFSET; else the position itself) and, with TWO, B = position + (b0, b1) go to acc.add(vA, vB, t) in the order of t.
// TRANSP: the tile's fast axis is the image's y axis and src is the transposed copy of the image (n_u rows of n_v
// texels); img is always the image itself (global-memory path of threads whose run fails the containment check).
// bf.sigma / bf.beta must describe the band; threads whose line does not run with it never use the tile.
// FMA: the contracted arithmetic convention (positions fmaf(t, d, o), ecc_bilerp<true>) instead of the exact one.
template <bool TRANSP, bool TWO, bool A_OFFSET, bool FMA, class Band, class Acc>
__device__ __forceinline__ void walk(Shared& sh, const float* __restrict__ img, int W, int H, const float* __restrict__ src,
                                     const BandFrame& bf, const Band& band, LineRun& ln, float a0, float a1, float b0, float b1,
                                     const float step, Acc& acc)
{
    const int tid = threadIdx.x;
    const int Nf = TRANSP ? H : W, Ns = TRANSP ? W : H;
    const float o0 = ln.o0, o1 = ln.o1, d0 = ln.d0, d1 = ln.d1, t_max = ln.t_max;
    float t = ln.t;
    // this thread in (f, s) coordinates
    const float of = TRANSP ? o1 : o0, os = TRANSP ? o0 : o1;
    const float df = TRANSP ? d1 : d0, ds = TRANSP ? d0 : d1;
    const float af = TRANSP ? a1 : a0, as = TRANSP ? a0 : a1;
    const float ef = TRANSP ? b1 : b0, es = TRANSP ? b0 : b1;
    const bool with_band = ln.active && ds * bf.sigma > .3f;
    const float inv_ds = with_band ? 1.f / ds : 0.f;

    // extent of the workgroup's samples along s
    float s_first, s_last;
    {
        const float sA = os + t * ds, sB = os + t_max * ds;
        s_first = with_band ? fminf(sA, sB) : FLT_MAX;
        s_last = with_band ? fmaxf(sA, sB) : -FLT_MAX;
        block_min_max(sh, s_first, s_last);
    }
    if (s_first <= s_last) {  // uniform: somebody samples from slabs
        const unsigned tile_addr = (unsigned)(size_t)(lds_v2f*)sh.tile;
        const float b_end = bf.sigma > 0.f ? s_last : s_first;
        stage_regs reg = {};
        SlabPlan nxt = plan_slab(bf, band, bf.sigma > 0.f ? s_first - .01f : s_last + .01f);
        stage_load(nxt, bf.bstep, src, Nf, Ns, reg);
        for (int it = 0; it < MAX_SLABS; ++it) {
            __syncthreads();  // everybody has left the tile
            stage_store(nxt, bf.bstep, tile_addr, reg);
            __syncthreads();  // slab complete
            const SlabPlan cur = nxt;
            const bool more = bf.sigma > 0.f ? cur.b_next <= b_end : cur.b_next >= b_end;  // uniform
            if (more) {
                nxt = plan_slab(bf, band, cur.b_next);
                stage_load(nxt, bf.bstep, src, Nf, Ns, reg);  // in flight while this slab is sampled
            }
            if (tid == 0) { ECC_SLAB_STAT(0, 1); ECC_SLAB_STAT(4, cur.S); ECC_SLAB_STAT(5, cur.H); }
            // samples of this slab: t <= lim (any monotone sequence of limits partitions the samples)
            const float t_end = fminf(t_max, (cur.b_next - os) * inv_ds);
            if (with_band && t <= t_end) {
                // Both end points of the run inside the slab's admissible region => every sample's footprint is in the
                // tile (the region is convex and positions are linear in t; .03 / .01 px cover the fp32 rounding of
                // o + t * d).  f, s >= .5 keeps the tile index non-negative; the guard carries a margin (GUARD = .5 + 1e-3)
                // because the end points are evaluated unfused here while the contracted mode samples at fmaf(t, d, o): a
                // fused position may round just below an unfused end point (a failing run only takes the global-memory
                // path -- same arithmetic, same bits).
                bool ok = true;
                constexpr float GUARD = .5f + 1e-3f;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float te = e ? t_end : t;
                    const float f = of + te * df, s = os + te * ds;
                    {
                        const float fa = A_OFFSET ? f + af : f, sa = A_OFFSET ? s + as : s;
                        const float g = fa - bf.beta * sa;
                        ok = ok && g >= cur.Glo && g <= cur.Ghi && sa >= cur.slo && sa <= cur.shi && fa >= GUARD && sa >= GUARD;
                    }
                    if (TWO) {
                        const float fb = f + ef, sb = s + es;
                        const float g = fb - bf.beta * sb;
                        ok = ok && g >= cur.Glo && g <= cur.Ghi && sb >= cur.slo && sb <= cur.shi && fb >= GUARD && sb >= GUARD;
                    }
                }
                if (ok) {
                    const float Sf = (float)cur.S;
                    const unsigned S8 = 8u * (unsigned)cur.S;
                    const unsigned base = tile_addr - 8u * (unsigned)(cur.R0 * cur.S + cur.I0) - (0x4B000000u << 3);
                    for (; t <= t_end; t += step) {  // t += step accumulates in fp32, like the reference's loops
                        ECC_SLAB_STAT(1, 1);
                        const float x = line_pos<FMA>(o0, t, d0), y = line_pos<FMA>(o1, t, d1);
                        const float vA = tex_pairs<TRANSP, FMA>(base, Sf, S8, A_OFFSET ? x + a0 : x, A_OFFSET ? y + a1 : y);
                        const float vB = TWO ? tex_pairs<TRANSP, FMA>(base, Sf, S8, x + b0, y + b1) : 0.f;
                        acc.add(vA, vB, t);
                    }
                } else {
                    for (; t <= t_end; t += step) {
                        ECC_SLAB_STAT(2, 1);
                        const float x = line_pos<FMA>(o0, t, d0), y = line_pos<FMA>(o1, t, d1);
                        const float vA = ecc_tex_global<FMA>(img, W, H, A_OFFSET ? x + a0 : x, A_OFFSET ? y + a1 : y);
                        const float vB = TWO ? ecc_tex_global<FMA>(img, W, H, x + b0, y + b1) : 0.f;
                        acc.add(vA, vB, t);
                    }
                }
            }
            if (!more) break;
        }
    }
    // Whatever the slabs did not cover: lines that do not run with the band, and the bound on the slab loop.
    if (ln.active)
        for (; t <= t_max; t += step) {
            ECC_SLAB_STAT(3, 1);
            const float x = line_pos<FMA>(o0, t, d0), y = line_pos<FMA>(o1, t, d1);
            const float vA = ecc_tex_global<FMA>(img, W, H, A_OFFSET ? x + a0 : x, A_OFFSET ? y + a1 : y);
            const float vB = TWO ? ecc_tex_global<FMA>(img, W, H, x + b0, y + b1) : 0.f;
            acc.add(vA, vB, t);
        }
    ln.t = t;
}

}  // namespace ecc_slab

#endif
