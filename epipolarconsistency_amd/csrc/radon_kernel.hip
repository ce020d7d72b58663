// radon_kernel.hip -- Radon intermediate (derivative / plain line integrals) for gfx950.
//
// Computes what the reference's radonDerivative<derivative> kernel computes
// (ref: LibEpipolarConsistency/RadonIntermediate.cu:32-143): one Radon bin per thread, stepping
// 0.66 px along the clipped line and accumulating bilinear samples SEQUENTIALLY in fp32 (the dtr
// is a difference of two large sums; re-associating them moves the result by >> 1e-5).
// Every float expression keeps the reference's source order and is compiled without contraction,
// so the result is bit-identical to oracle/ecc_oracle.c (or_radon_bin).
//
// What is different from the reference is the machine mapping (no texture unit; the bilinear filter is the exact
// fp32 rule of SURVEY.md 8c, evaluated from LDS):
//   * a workgroup owns 16 adjacent angles x 16 adjacent distances (256 threads).  Its lines are nearly parallel, so
//     they sweep a narrow band of the image.  The band is cut into SLABS across the image axis the lines run along
//     (the "slow" axis s; the other one, closer to the line normal, is the "fast" axis f); every thread walks its
//     own t-loop through the slab, so each bin still visits its samples in the reference's order;
//   * the slab lives in LDS as TEXEL PAIRS: element (i, r) = {T(i, r), T(i+1, r)} (float2, clamp addressing
//     resolved while staging), so one aligned ds_read_b64 fetches a footprint row: 2 x 2 LDS cycles per bilinear
//     sample where 2 x ds_read2_b32 took 2 x 8 (the two angles that share a 32-lane group are 2-way conflicting
//     in either form) -- the kernel is bound by vector-ALU issue instead of the LDS pipe;
//   * the tile is a LINEAR array, address(i, r) = (r - R0) * S + (i - I0), with the row stride S a multiple of 32
//     pairs (bank = i mod 32, whatever the row).  Row r only ever holds the S - 1 columns from
//     ws(r) = floor(alpha + beta * r) on, beta = the band's slope df/ds: the window slides with the band, so a
//     slab stores a parallelogram, not the band's bounding box, and S only has to cover the band's width ALONG f
//     (64 ... 256, chosen per slab).  A slab is floor(5056 / (S + 1)) rows thick whatever the angle;
//   * slab geometry is analytic (two corner angles x two corner distances of the workgroup) and identical in
//     every thread: no reductions, no decisions through LDS.  It does not have to be trusted: every thread checks
//     the two end points of its own run through the slab against the slab's admissible region (a convex set:
//     two half-planes in the sheared coordinate f - beta*s, two in s) and samples from global memory when the
//     check fails (weird proportions only, e.g. 5 angle bins for a 300-pixel image);
//   * the next slab's texels are loaded into registers BEFORE the current slab is sampled and stored to LDS
//     after it: two barriers per slab and no exposed global-memory latency;
//   * the index conversion float -> LDS address is one fp32 add of 2^23 and a shift-add on the bits instead of
//     v_cvt_i32_f32 (quarter rate);
//   * sin/cos of the bin angle come from a host table (one entry per angle), not per thread;
//   * output goes to the transposed, border-padded layout of ecc_layout.h (distance fastest).
#include <hip/hip_runtime.h>
#include <float.h>
#include <limits.h>

#include "ecc_layout.h"
#include "ecc_sampling.h"

#ifdef ECC_RADON_STATS
// 0: slabs, 1: LDS-path steps, 2: global-path steps inside slabs, 3: steps of the safety net, 4: sum of S, 5: sum of H
__device__ unsigned long long g_radon_stats[8];
#define RSTAT(i, v) atomicAdd(&g_radon_stats[i], (unsigned long long)(v))
#else
#define RSTAT(i, v)
#endif

namespace {

constexpr int RT_T = 16;                   // distance bins per workgroup
constexpr int RT_A = 16;                   // angle bins per workgroup
constexpr int RT_THREADS = RT_T * RT_A;    // 256
constexpr int RT_WAVES = RT_THREADS / 64;
constexpr int TILE_CAP = 5056;             // texel pairs per workgroup: 40 448 B, four workgroups per CU
constexpr int N_PRE = 26;                  // texels a thread stages per slab (registers that live across the sampling loop)
static_assert(N_PRE <= 32, "stage_regs is a 32-float vector");
constexpr int S_MIN = 64, S_MAX = 256;     // row stride of the tile in pairs (multiples of 32)
constexpr float RADON_STEP = .66f;         // ref: RadonIntermediate.cu:102
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

struct RadonShared {
    float2 tile[TILE_CAP];
    float srange[RT_WAVES][2];
};

// The workgroup's band, in (f, s) coordinates (all values identical in every thread).
// A sample of the line with normal (l0, l1) and offset c (position = c * normal + (.5, .5) + t * direction) at slow
// coordinate s has fast coordinate f = .5 + (s - .5) * m + c / lf, with lf the normal's f-component and m = -ls / lf.
struct Band {
    float beta, abs_beta;  // slope df/ds the row windows follow (middle angle, clamped to [-1, 1])
    int bstep;             // beta in 1/65536 columns per row: the row windows are placed by integer arithmetic
    float sigma;           // +1: s grows with t, -1: s falls
    float mk[2];           // m - beta of the first / last angle
    float g0[2];           // .5 - .5 m
    float cl[2], ch[2];    // c / lf at the workgroup's smallest / largest offset (either order)
    float marg;            // allowance for the angles in between and for rounding
};

struct SlabPlan {
    int R0, H, S, I0;      // first staged row, rows, row stride (pairs), column of address 0
    int acc0;              // row window of tile row rr: ws = (acc0 + rr * bstep) >> 16 = floor(alpha + beta * (R0 + rr)) up to
                           // 1e-3 columns, S - 1 elements from there
    float Glo, Ghi;        // admissible f - beta * s of a sample
    float slo, shi;        // admissible s of a sample
    float b_next;          // samples with s up to b_next (in walking order) belong to this slab
};

// values every lane computes identically, moved to scalar registers (branches on them become scalar branches)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// Rows of a slab by k = S / 32: the linear tile needs H * (S + 1) pairs (the row windows slide by up to one column per
// row); wave w stages rows w, w + 4, ..., a row in ceil(S / 64) loads of 64 consecutive texels, N_PRE registers in all.
constexpr int slab_rows_of(int k)
{
    return (TILE_CAP / (32 * k + 1)) < RT_WAVES * (N_PRE / ((k + 1) / 2)) ? (TILE_CAP / (32 * k + 1)) : RT_WAVES * (N_PRE / ((k + 1) / 2));
}
constexpr unsigned long long SLAB_ROWS_PACKED = (unsigned long long)slab_rows_of(2) | ((unsigned long long)slab_rows_of(3) << 8) |
                                                ((unsigned long long)slab_rows_of(4) << 16) | ((unsigned long long)slab_rows_of(5) << 24) |
                                                ((unsigned long long)slab_rows_of(6) << 32) | ((unsigned long long)slab_rows_of(7) << 40) |
                                                ((unsigned long long)slab_rows_of(8) << 48);
__device__ __forceinline__ int slab_rows(int k) { return (int)((SLAB_ROWS_PACKED >> (8 * (k - 2))) & 255ull); }
static_assert(S_MIN == 64 && S_MAX == 256 && slab_rows_of(2) < 256, "slab_rows covers k = 2 .. 8");

// Slab that starts at slow coordinate b (walking order).  Every thread computes the same plan.
__device__ __forceinline__ SlabPlan plan_slab(const Band& bd, float b)
{
    SlabPlan sp;
    constexpr int HMAX = slab_rows_of(S_MIN / 32);
    const float span = (float)(HMAX - 5);
    const float sa = (bd.sigma > 0.f ? b : b - span) - 1.f, sb = sa + span + 2.f;
    float gmin = FLT_MAX, gmax = -FLT_MAX;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float u0 = bd.mk[e] * sa, u1 = bd.mk[e] * sb;
        gmin = fminf(gmin, bd.g0[e] + fminf(u0, u1) + fminf(bd.cl[e], bd.ch[e]));
        gmax = fmaxf(gmax, bd.g0[e] + fmaxf(u0, u1) + fmaxf(bd.cl[e], bd.ch[e]));
    }
    const float need = (gmax - gmin) + 2.f * bd.marg + 2.f + 3.f * bd.abs_beta + .25f;
    const int S = uni(min(max(((int)ceilf(fminf(need, 4096.f)) + 31) & ~31, S_MIN), S_MAX));
    const int H = slab_rows(S >> 5);
    const float delta = (float)(H - 5);
    const float slack = (float)S - need;  // negative: the band does not fit, the per-thread checks decide
    sp.S = S;
    sp.H = H;
    const float alpha = gmin - bd.marg - .5f - 1.5f * bd.abs_beta - .125f - .5f * slack;
    sp.Glo = alpha + .5f + 1.5f * bd.abs_beta + .03f;
    sp.Ghi = alpha + (float)S - 1.5f - 1.5f * bd.abs_beta - .03f;
    const float lo = bd.sigma > 0.f ? b : b - delta;
    sp.R0 = uni((int)floorf(lo - 1.75f));
    sp.slo = (float)sp.R0 + .51f;
    sp.shi = (float)(sp.R0 + H) - .51f;
    sp.b_next = uni(b + bd.sigma * delta);
    sp.acc0 = uni((int)floorf(__builtin_fmaf(bd.beta, (float)sp.R0, alpha) * 65536.f));
    sp.I0 = min(sp.acc0 >> 16, (sp.acc0 + (H - 1) * bd.bstep) >> 16);
    return sp;
}

// Staging, first half: this thread's texels of the slab into registers.  Wave w owns rows w, w + 4, ...; a row is
// K = ceil(S / 64) loads of 64 consecutive texels (256 contiguous bytes per load); register j * K + seg holds load seg of
// the wave's j-th row.  Everything but the lane's column is wave-uniform and stays in scalar registers.  src is the
// image with f as its fast axis (the image itself, or its transposed copy), Nf x Ns texels.
// rows a wave stages when a row takes K loads: ceil(rows of the smallest such stride / 4)
constexpr int stage_rows_of(int K) { return (slab_rows_of(K == 1 ? 2 : 2 * K - 1) + RT_WAVES - 1) / RT_WAVES; }
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
        acc += RT_WAVES * bstep;
        r += RT_WAVES;
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
        acc += RT_WAVES * bstep;
        rr += RT_WAVES;
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
template <bool TRANSP>
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
    const v2f pa = *(const lds_v2f*)(size_t)a0;
    const v2f pb = *(const lds_v2f*)(size_t)a1;
    const float T00 = pa.x, T10 = TRANSP ? pb.x : pa.y;
    const float T01 = TRANSP ? pa.y : pb.x, T11 = pb.y;
    float r0 = (1.f - fx) * T00 + fx * T10;
    float r1 = (1.f - fx) * T01 + fx * T11;
    return (1.f - fy) * r0 + fy * r1;
}

template <bool DERIV, bool TRANSP>
__device__ __forceinline__ void radon_body(const EccRadonParams& p, RadonShared& sh)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    // a 32-lane group (what an LDS read is serviced in) is 16 adjacent distances x 2 adjacent angles
    const int ix = blockIdx.x * RT_A + ((tid >> 4) & (RT_A - 1));
    const int iy = blockIdx.y * RT_T + (tid & 15);
    const float* __restrict__ img = p.images + (int64_t)blockIdx.z * p.image_stride;
    const int W = p.n_u, H = p.n_v;
    const float n_u = (float)W, n_v = (float)H;
    const bool in_range = ix < p.n_alpha && iy < p.n_t;

    // ---- per-bin line set-up, ref: RadonIntermediate.cu:36-99 (same expressions, same order) ----
    float o0 = 0.f, o1 = 0.f, d0 = 1.f, d1 = 0.f, t = 0.f, t_max = -1.f;
    bool active = false;
    const float diag = sqrtf(n_u * n_u + n_v * n_v);
    if (in_range) {
        float y_rel = (iy / (float)p.n_t - 0.5f);
        float tau = y_rel * diag;
        float l0 = -p.trig[2 * ix];
        float l1 = p.trig[2 * ix + 1];
        float l2 = -tau;
        l2 += -0.5f * n_u * l0 - 0.5f * n_v * l1;
        o0 = -l2 * l0;
        o1 = -l2 * l1;
        d0 = l1;
        d1 = -l0;
        float ts[4] = {(1.f - o0) / d0, (n_u - 1.f - o0) / d0, (1.f - o1) / d1, (n_v - 1.f - o1) / d1};
        if (d0 * d0 < 1e-12f) ts[0] = -(ts[1] = 1e10f);
        if (d1 * d1 < 1e-12f) ts[2] = -(ts[3] = 1e10f);
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int i = 0; i < 3; i++)
                if (ts[i] > ts[i + 1]) {
                    float tmp = ts[i];
                    ts[i] = ts[i + 1];
                    ts[i + 1] = tmp;
                }
        t = ts[1];
        t_max = ts[2];
        float u = o0 + t * d0, v = o1 + t * d1;
        bool inb = (u <= n_u && v <= n_v && u >= 0 && v >= 0);
        active = inb && !(t_max <= t);
        o0 += .5f;
        o1 += .5f;
        if (DERIV) {
            o0 -= .5f * d1;
            o1 += .5f * d0;
        }
    }

    // ---- the workgroup's band (identical in every thread) ----
    const int Nf = TRANSP ? H : W, Ns = TRANSP ? W : H;
    const float* __restrict__ src = TRANSP ? (p.imagesT + (int64_t)blockIdx.z * p.image_stride) : img;
    const int ixA = blockIdx.x * RT_A, ixB = min(ixA + RT_A - 1, p.n_alpha - 1), ixM = (ixA + ixB) >> 1;
    const int iyA = blockIdx.y * RT_T, iyB = min(iyA + RT_T - 1, p.n_t - 1);
    Band bd;
    {
        const float lfM = TRANSP ? p.trig[2 * ixM + 1] : -p.trig[2 * ixM];
        const float lsM = TRANSP ? -p.trig[2 * ixM] : p.trig[2 * ixM + 1];
        // direction (d0, d1) = (l1, -l0): its s-component is -l0 (plain) or l1 (transposed)
        const float dsM = TRANSP ? lfM : -lfM;
        bd.sigma = dsM >= 0.f ? 1.f : -1.f;
        bd.beta = uni(fminf(fmaxf(-lsM / lfM, -1.f), 1.f));
        bd.abs_beta = fabsf(bd.beta);
        bd.bstep = uni((int)rintf(bd.beta * 65536.f));
        const float tauA = ((float)iyA / (float)p.n_t - 0.5f) * diag, tauB = ((float)iyB / (float)p.n_t - 0.5f) * diag;
        const float Pi = 3.14159265359f;
        const float da = (float)(ixB - ixA) * (Pi / (float)p.n_alpha);
        bd.marg = uni(.75f + .6f * diag * da * da);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int ixe = e ? ixB : ixA;
            const float l0 = -p.trig[2 * ixe], l1 = p.trig[2 * ixe + 1];
            const float lf = TRANSP ? l1 : l0, ls = TRANSP ? l0 : l1;
            const float inv = 1.f / lf, m = -ls * inv;
            const float kc = 0.5f * n_u * l0 + 0.5f * n_v * l1;
            bd.mk[e] = uni(m - bd.beta);
            bd.g0[e] = uni(.5f - .5f * m);
            bd.cl[e] = uni((tauA + kc - .5f) * inv);
            bd.ch[e] = uni((tauB + kc + .5f) * inv);
        }
    }

    // ---- this thread in (f, s) coordinates ----
    const float of = TRANSP ? o1 : o0, os = TRANSP ? o0 : o1;
    const float df = TRANSP ? d1 : d0, ds = TRANSP ? d0 : d1;
    const float ef = TRANSP ? -d0 : d1, es = TRANSP ? d1 : -d0;  // second sample of the derivative pair: (x + d1, y - d0)
    // threads whose line does not run with the band (few angle bins for the image size) never use the tile
    const bool with_band = active && ds * bd.sigma > .3f;
    const float inv_ds = with_band ? 1.f / ds : 0.f;

    // ---- extent of the workgroup's samples along s ----
    {
        const float sA = os + t * ds, sB = os + t_max * ds;
        float smin = with_band ? fminf(sA, sB) : FLT_MAX;
        float smax = with_band ? fmaxf(sA, sB) : -FLT_MAX;
        smin = wave_min_f(smin);
        smax = wave_max_f(smax);
        if (lane == 0) {
            sh.srange[wave][0] = smin;
            sh.srange[wave][1] = smax;
        }
    }
    __syncthreads();
    float s_first = sh.srange[0][0], s_last = sh.srange[0][1];
#pragma unroll
    for (int q = 1; q < RT_WAVES; ++q) {
        s_first = fminf(s_first, sh.srange[q][0]);
        s_last = fmaxf(s_last, sh.srange[q][1]);
    }
    s_first = uni(s_first);
    s_last = uni(s_last);

    float sum = 0.f, sumo = 0.f;
    if (s_first <= s_last) {  // uniform: somebody samples from slabs
        const unsigned tile_addr = (unsigned)(size_t)(lds_v2f*)sh.tile;
        const float b_end = bd.sigma > 0.f ? s_last : s_first;
        stage_regs reg = {};
        SlabPlan nxt = plan_slab(bd, bd.sigma > 0.f ? s_first - .01f : s_last + .01f);
        stage_load(nxt, bd.bstep, src, Nf, Ns, reg);
        for (int it = 0; it < MAX_SLABS; ++it) {
            __syncthreads();  // everybody has left the tile
            stage_store(nxt, bd.bstep, tile_addr, reg);
            __syncthreads();  // slab complete
            const SlabPlan cur = nxt;
            const bool more = bd.sigma > 0.f ? cur.b_next <= b_end : cur.b_next >= b_end;  // uniform
            if (more) {
                nxt = plan_slab(bd, cur.b_next);
                stage_load(nxt, bd.bstep, src, Nf, Ns, reg);  // in flight while this slab is sampled
            }
            if (tid == 0) { RSTAT(0, 1); RSTAT(4, cur.S); RSTAT(5, cur.H); }
            // samples of this slab: t <= lim (any monotone sequence of limits partitions the samples)
            const float t_end = fminf(t_max, (cur.b_next - os) * inv_ds);
            if (with_band && t <= t_end) {
                // Both end points of the run inside the slab's admissible region => every sample's footprint is in the
                // tile (the region is convex and positions are linear in t; .03 / .01 px cover the fp32 rounding of
                // o + t * d).  f, s >= .5 keeps the tile index non-negative.
                bool ok = true;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float te = e ? t_end : t;
                    float f = of + te * df, s = os + te * ds;
                    float g = f - bd.beta * s;
                    ok = ok && g >= cur.Glo && g <= cur.Ghi && s >= cur.slo && s <= cur.shi && f >= .5f && s >= .5f;
                    if (DERIV) {
                        f += ef;
                        s += es;
                        g = f - bd.beta * s;
                        ok = ok && g >= cur.Glo && g <= cur.Ghi && s >= cur.slo && s <= cur.shi && f >= .5f && s >= .5f;
                    }
                }
                if (ok) {
                    const float Sf = (float)cur.S;
                    const unsigned S8 = 8u * (unsigned)cur.S;
                    const unsigned base = tile_addr - 8u * (unsigned)(cur.R0 * cur.S + cur.I0) - (0x4B000000u << 3);
                    // ref: RadonIntermediate.cu:105-123 (t += step accumulates in fp32)
                    for (; t <= t_end; t += RADON_STEP) {
                        RSTAT(1, 1);
                        float x = o0 + t * d0, y = o1 + t * d1;
                        sum += tex_pairs<TRANSP>(base, Sf, S8, x, y);
                        if (DERIV) sumo += tex_pairs<TRANSP>(base, Sf, S8, x + d1, y - d0);
                    }
                } else {
                    for (; t <= t_end; t += RADON_STEP) {
                        RSTAT(2, 1);
                        float x = o0 + t * d0, y = o1 + t * d1;
                        sum += ecc_tex_global(img, W, H, x, y);
                        if (DERIV) sumo += ecc_tex_global(img, W, H, x + d1, y - d0);
                    }
                }
            }
            if (!more) break;
        }
    }
    // Whatever the slabs did not cover: lines that do not run with the band, and the bound on the slab loop.
    if (active)
        for (; t <= t_max; t += RADON_STEP) {
            RSTAT(3, 1);
            float x = o0 + t * d0, y = o1 + t * d1;
            sum += ecc_tex_global(img, W, H, x, y);
            if (DERIV) sumo += ecc_tex_global(img, W, H, x + d1, y - d0);
        }

    if (in_range) {
        float result;
        if (!active) result = 0.f;
        else if (!DERIV) result = sum * RADON_STEP;
        else {
            result = (sum - sumo) * RADON_STEP;  // ref: RadonIntermediate.cu:125-140
            if (p.post_process == 1) result = result < 0 ? -sqrtf(-result) : sqrtf(result);
            // logarithm correctly rounded (binary64, rounded once) like the oracle's: once per bin
            else if (p.post_process == 2) result = result < 0 ? -(float)log((double)(-result + 1)) : (float)log((double)(result + 1));
        }
        float* out = p.out + (int64_t)blockIdx.z * p.out_stride;
        out[(size_t)(ix + 1) * p.pitch + (iy + 1)] = result;
    }
}

template <bool DERIV>
__global__ __launch_bounds__(RT_THREADS) void radon_kernel(EccRadonParams p)
{
    __shared__ RadonShared sh;
    // Line normal of the workgroup's middle angle: (nx, ny) = (-sin a, cos a).  The tile's fast axis is the image
    // axis the normal is closer to: a 32-lane group is 16 adjacent distances x 2 adjacent angles, i.e. two nearly
    // coincident "combs" of 16 points spaced 1.9 px along the normal; with the bank = fast index mod 32 a comb
    // advances >= 1.33 banks per lane over < 32 banks, conflict-free in itself, and the two combs cost exactly two
    // passes (scripts/analysis/radon_lds_layouts.py: no bank function of (x, y) separates them).  Normals closer
    // to y read a transposed copy of the image stack (made by ecc_launch_radon's caller), so staging stays coalesced.
    const int ixA = blockIdx.x * RT_A, ixB = min(ixA + RT_A - 1, p.n_alpha - 1), ixM = (ixA + ixB) >> 1;
    const float nx = -p.trig[2 * ixM], ny = p.trig[2 * ixM + 1];
    if (fabsf(nx) >= fabsf(ny))
        radon_body<DERIV, false>(p, sh);
    else
        radon_body<DERIV, true>(p, sh);
}

// Replicate the border rows/columns of the private layout (clamp addressing, ecc_layout.h).
__global__ void dtr_border_kernel(float* slabs, int64_t stride, int n_alpha, int n_t, int pitch)
{
    float* s = slabs + (int64_t)blockIdx.z * stride;
    const int rows = n_alpha + 2, cols = n_t + 2;
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    int R, C;
    if (e < cols) { R = 0; C = e; }
    else if (e < 2 * cols) { R = rows - 1; C = e - cols; }
    else if (e < 2 * cols + rows) { R = e - 2 * cols; C = 0; }
    else if (e < 2 * cols + 2 * rows) { R = e - 2 * cols - rows; C = cols - 1; }
    else return;
    int r = min(max(R, 1), n_alpha), c = min(max(C, 1), n_t);
    s[(size_t)R * pitch + C] = s[(size_t)r * pitch + c];
}

// alpha-fast (API) <-> private layout conversion, tiled transpose through LDS.
__global__ void dtr_import_kernel(const float* __restrict__ src, float* __restrict__ slab, int n_alpha,
                                  int n_t, int pitch)
{
    __shared__ float tl[32][33];
    int ax = blockIdx.x * 32, ty = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        int iy = ty + r, ix = ax + threadIdx.x;
        tl[r][threadIdx.x] = (iy < n_t && ix < n_alpha) ? src[(size_t)iy * n_alpha + ix] : 0.f;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        int ix = ax + r, iy = ty + threadIdx.x;
        if (ix < n_alpha && iy < n_t) slab[(size_t)(ix + 1) * pitch + (iy + 1)] = tl[threadIdx.x][r];
    }
}

__global__ void dtr_export_kernel(const float* __restrict__ slab, float* __restrict__ dst, int n_alpha,
                                  int n_t, int pitch)
{
    __shared__ float tl[32][33];
    int ax = blockIdx.x * 32, ty = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        int ix = ax + r, iy = ty + threadIdx.x;
        tl[r][threadIdx.x] = (ix < n_alpha && iy < n_t) ? slab[(size_t)(ix + 1) * pitch + (iy + 1)] : 0.f;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        int iy = ty + r, ix = ax + threadIdx.x;
        if (iy < n_t && ix < n_alpha) dst[(size_t)iy * n_alpha + ix] = tl[threadIdx.x][r];
    }
}

}  // namespace

// ---- launchers (host) --------------------------------------------------------------------------
extern "C" hipError_t ecc_launch_radon(const EccRadonParams* p, int derivative, hipStream_t stream)
{
    if (!p->images || !p->imagesT || !p->out || !p->trig) return hipErrorInvalidValue;  // both image copies are read
    dim3 grid((p->n_alpha + RT_A - 1) / RT_A, (p->n_t + RT_T - 1) / RT_T, p->n_img);
    dim3 block(RT_THREADS);
    if (derivative)
        hipLaunchKernelGGL(radon_kernel<true>, grid, block, 0, stream, *p);
    else
        hipLaunchKernelGGL(radon_kernel<false>, grid, block, 0, stream, *p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    int border = 2 * (p->n_t + 2) + 2 * (p->n_alpha + 2);
    hipLaunchKernelGGL(dtr_border_kernel, dim3((border + 255) / 256, 1, p->n_img), dim3(256), 0, stream,
                       p->out, p->out_stride, p->n_alpha, p->n_t, p->pitch);
    return hipGetLastError();
}

extern "C" hipError_t ecc_launch_dtr_border(float* slabs, int64_t slab_stride, int n_img, int n_alpha, int n_t,
                                            int pitch, hipStream_t stream)
{
    int border = 2 * (n_t + 2) + 2 * (n_alpha + 2);
    hipLaunchKernelGGL(dtr_border_kernel, dim3((border + 255) / 256, 1, n_img), dim3(256), 0, stream, slabs,
                       slab_stride, n_alpha, n_t, pitch);
    return hipGetLastError();
}

extern "C" hipError_t ecc_launch_dtr_import(const float* src_alpha_fast, float* slab, int n_alpha, int n_t,
                                            int pitch, hipStream_t stream)
{
    dim3 grid((n_alpha + 31) / 32, (n_t + 31) / 32), block(32, 8);
    hipLaunchKernelGGL(dtr_import_kernel, grid, block, 0, stream, src_alpha_fast, slab, n_alpha, n_t, pitch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    int border = 2 * (n_t + 2) + 2 * (n_alpha + 2);
    hipLaunchKernelGGL(dtr_border_kernel, dim3((border + 255) / 256, 1, 1), dim3(256), 0, stream, slab,
                       (int64_t)0, n_alpha, n_t, pitch);
    return hipGetLastError();
}

extern "C" hipError_t ecc_launch_dtr_export(const float* slab, float* dst_alpha_fast, int n_alpha, int n_t,
                                            int pitch, hipStream_t stream)
{
    dim3 grid((n_alpha + 31) / 32, (n_t + 31) / 32), block(32, 8);
    hipLaunchKernelGGL(dtr_export_kernel, grid, block, 0, stream, slab, dst_alpha_fast, n_alpha, n_t, pitch);
    return hipGetLastError();
}

#ifdef ECC_RADON_STATS
extern "C" __attribute__((visibility("default"))) void ecc_debug_radon_stats(unsigned long long* out, int reset)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_radon_stats), sizeof(unsigned long long) * 8);
    if (reset) {
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_radon_stats), z, sizeof(z));
    }
}
#endif
