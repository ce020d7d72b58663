// radon_kernel.hip -- Radon intermediate (derivative / plain line integrals) for gfx950.
//
// Computes what the reference's radonDerivative<derivative> kernel computes
// (ref: LibEpipolarConsistency/RadonIntermediate.cu:32-143): one Radon bin per thread, stepping
// 0.66 px along the clipped line and accumulating bilinear samples SEQUENTIALLY in fp32 (the dtr
// is a difference of two large sums; re-associating them moves the result by >> 1e-5).
// Every float expression keeps the reference's source order and is compiled without contraction,
// so the result is bit-identical to oracle/ecc_oracle.c (or_radon_bin).
//
// What is different from the reference is the machine mapping:
//   * no texture unit: the bilinear filter is the exact fp32 rule of SURVEY.md 8c, evaluated from
//     an LDS tile (4 taps = 2 x ds_read2_b32) instead of 4 scattered global loads; built without the SLP
//     vectoriser (build.py): v_pk_*_f32 pairs cost two issue slots each on gfx950 plus operand shuffles;
//   * a workgroup owns RT_A = 16 adjacent angles x RT_T = 16 adjacent distances (256 threads).  Its lines are
//     (nearly) parallel, so they sweep a narrow band of the image.  The band is walked in chunks
//     along the line direction; for every chunk the axis-aligned bounding box of all sample
//     footprints is staged into LDS with coalesced row reads (border replicated = clamp addressing),
//     then every thread advances its own t-loop through the chunk.  Each thread still visits its
//     samples in the reference's order, so sums are unchanged;
//   * sin/cos of the bin angle come from a host table (one entry per angle), not per thread;
//   * output goes to the transposed, border-padded layout of ecc_layout.h (distance fastest), so
//     the 32 lanes of a half-wave write one 128-B segment.
#include <hip/hip_runtime.h>
#include <float.h>
#include <limits.h>

#include "ecc_layout.h"
#include "ecc_sampling.h"

#ifdef ECC_RADON_STATS
__device__ unsigned long long g_radon_stats[8];
#define RSTAT(i, v) atomicAdd(&g_radon_stats[i], (unsigned long long)(v))
#else
#define RSTAT(i, v)
#endif

namespace {

#ifndef RT_DIST_BINS
#define RT_DIST_BINS 16
#endif
#ifndef RT_WG_THREADS
#define RT_WG_THREADS 256
#endif
constexpr int RT_T = RT_DIST_BINS;                  // distance bins per workgroup (tid % RT_T)
constexpr int RT_A = RT_WG_THREADS / RT_DIST_BINS;  // angle bins per workgroup    (tid / RT_T)
constexpr int RT_THREADS = RT_T * RT_A;             // 256 (512 only in experiments)
constexpr int RT_WAVES = RT_THREADS / 64;
constexpr int RT_STAGE_ROWS = RT_THREADS / 32;      // tile rows staged per pass (32 texels per row segment)
// Tile shape: 96 x 96 floats (37 KB, 4 workgroups per CU) measured best; -DRT_TILE_W/H only for experiments
// (scripts/radon_variants.sh: 64x64 0.97 ms, 96x80 0.79 ms, 96x96 0.76 ms per 1024^2 image).
#ifndef RT_TILE_W
#define RT_TILE_W 96
#endif
#ifndef RT_TILE_H
#define RT_TILE_H 96
#endif
constexpr int TILE_W = RT_TILE_W;          // usable LDS tile width (texels)
constexpr int TILE_H = RT_TILE_H;          // LDS tile rows
constexpr int TILE_S_MAX = TILE_W + 1;     // row stride is TILE_W+1 or TILE_W-1 floats (odd, see radon_kernel)
constexpr float RADON_STEP = .66f;         // ref: RadonIntermediate.cu:102
constexpr int MAX_CHUNKS = 8192;           // bound on the chunk loop (every spin is bounded)

// Exact fp32 bilinear rule on global memory with clamp addressing (ecc_sampling.h); slow path used
// only when a chunk's footprint cannot be made to fit the LDS tile.
__device__ __forceinline__ float tex_global(const float* __restrict__ img, int W, int H, float x, float y)
{
    return ecc_tex_global(img, W, H, x, y);
}

// Same rule on the staged tile.  tile_off = by0*TILE_S + bx0 (tile origin in image texels); the
// tile already holds clamped (replicated) texels, so taps need no index clamps.
// TRANSP: the tile holds the image transposed (image y is the tile's fast axis; staged from a transposed copy of the image).
template <int TILE_S, bool TRANSP>
__device__ __forceinline__ float tex_lds(const float* tile_shifted, float x, float y)
{
    float xb = x - 0.5f, yb = y - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    float fx = xb - fi, fy = yb - fj;
    // fj*TILE_S + fi is an exact small integer in fp32 (|.| < 2^24) whether or not it is fused;
    // tile_shifted = tile - tile_off folds the tile origin into the base (one v_lshl_add per sample).
    // (an explicit fma: the build runs with -ffp-contract=off, and product and sum are exact integers either way)
    const float* tp = tile_shifted + (TRANSP ? (int)__builtin_fmaf(fi, (float)TILE_S, fj) : (int)__builtin_fmaf(fj, (float)TILE_S, fi));
    float T00 = tp[0], T10 = tp[TRANSP ? TILE_S : 1];
    float T01 = tp[TRANSP ? 1 : TILE_S], T11 = tp[TILE_S + 1];
    float r0 = (1.f - fx) * T00 + fx * T10;
    float r1 = (1.f - fx) * T01 + fx * T11;
    return (1.f - fy) * r0 + fy * r1;
}

__device__ __forceinline__ int wave_min_i(int v)
{
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    return v;
}
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
    float tile[TILE_S_MAX * TILE_H];
    int box[RT_WAVES][4];    // per wave: min x, min y, max x, max y
    float u[RT_WAVES][2];    // per wave: min/max of the along-line coordinate
    float geo[6];
    int pend[RT_WAVES];
};

// TILE_S is the LDS row stride.  A half-wave is 16 adjacent distance bins x 2 adjacent angles, i.e.
// sample points spaced 1.9 px along the line NORMAL (nx, ny).  ds_read2_b32 banks are
// (j*TILE_S + i) mod 32: with stride 97 the bank advances by 1.9 (nx + ny) per lane, with 95 by
// 1.9 (nx - ny); the kernel picks the one with the larger advance, so a half-wave never walks along an
// iso-bank direction.  Measured (profiles/): the LDS pipe is busy ~90 % of the kernel time and bank
// conflicts are ~55 % of those cycles -- 32 lanes spread over >= 30 px of a line cannot all land on distinct
// banks of a linear layout.  Tried and rejected: a float2 "texel pair" tile read with ds_read_b64 (256 B/clk,
// 64 banks) halves the LDS cycles but doubles the tile to 74 KB = 2 workgroups per CU, and this kernel needs
// >= 4 waves per SIMD to keep the VALU fed (2 WG/CU: 0.96 ms with the plain tile, 1.12 ms with pairs).
template <bool DERIV, int TILE_S, bool TRANSP = false>
__device__ __forceinline__ void radon_body(const EccRadonParams& p, RadonShared& sh)
{
    float* tile = sh.tile;
    auto& s_box = sh.box;
    auto& s_u = sh.u;
    auto& s_pend = sh.pend;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int ix = blockIdx.x * RT_A + (tid / RT_T);
    const int iy = blockIdx.y * RT_T + (tid % RT_T);
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

    // ---- chunking coordinate: position along the line direction relative to the image centre ----
    const float tc = (0.5f * n_u) * d0 + (0.5f * n_v) * d1;
    {
        float umin = active ? t - tc : FLT_MAX;
        float umax = active ? t_max - tc : -FLT_MAX;
        umin = wave_min_f(umin);
        umax = wave_max_f(umax);
        if (lane == 0) {
            s_u[wave][0] = umin;
            s_u[wave][1] = umax;
        }
    }
    __syncthreads();
    float U0 = s_u[0][0], U1 = s_u[0][1];
#pragma unroll
    for (int q = 1; q < RT_WAVES; ++q) {
        U0 = fminf(U0, s_u[q][0]);
        U1 = fmaxf(U1, s_u[q][1]);
    }
    if (tid == 0) {
        // Geometry of this workgroup's band for the chunk-length rule below (uniform via LDS).
        const float Pi = 3.14159265359f;
        const float dtau = diag / (float)p.n_t;
        const int iy0 = blockIdx.y * RT_T;
        const float tau_a = fabsf(((float)iy0 / (float)p.n_t - 0.5f) * diag);
        const float tau_b = fabsf(((float)(iy0 + RT_T - 1) / (float)p.n_t - 0.5f) * diag);
        sh.geo[0] = fabsf(d0);                                    // |cos| of the line direction
        sh.geo[1] = fabsf(d1);                                    // |sin|
        sh.geo[2] = (RT_T - 1) * dtau + 2.f;                      // band across the lines (+ derivative pair)
        sh.geo[3] = (RT_A - 1) * (Pi / (float)p.n_alpha);         // angular spread of the workgroup
        sh.geo[4] = fmaxf(tau_a, tau_b);                          // largest |distance to centre|
        sh.geo[5] = fmaxf(fabsf(U0), fabsf(U1));                  // largest |along-line coordinate|
    }
    __syncthreads();
    const float g_cs = sh.geo[0], g_sn = sh.geo[1], g_band = sh.geo[2], g_spread = sh.geo[3];
    const float g_tau = sh.geo[4], g_far = sh.geo[5];

    float sum = 0.f, sumo = 0.f;
    float U = U0;
    for (int it = 0; it < MAX_CHUNKS; ++it) {
        // Chunk length L: the chunk covers along-line coordinates [U, U+L) of every line of the workgroup.
        // Seen from the image centre the 8 angles rotate the (L x band) rectangle by up to `spread`, which
        // widens it by spread*|u| across the lines and lengthens it by spread*|tau| along them; L is the
        // largest length whose axis-aligned bounding box still fits the LDS tile (identical in all threads).
        const float reach = fminf(g_far, fabsf(U) + (float)TILE_W);
        const float band = g_band + g_spread * reach;
        const float slack = g_spread * g_tau;
        const float Lw = g_cs > 1e-3f ? ((float)(TILE_W - 3) - band * g_sn) / g_cs : 1e9f;
        const float Lh = g_sn > 1e-3f ? ((float)(TILE_H - 3) - band * g_cs) / g_sn : 1e9f;
        float L = fminf(fmaxf(fminf(Lw, Lh) - slack, 4.f), 4096.f);
        const bool pending = active && (t <= t_max);  // samples left at all
        const unsigned long long pend = __ballot(pending);
        // The rule above is an estimate; the exact footprint decides.  If it does not fit, the chunk is halved
        // (uniform decision, at most 4 times) rather than sampled from global memory, which costs ~200x per
        // sample.
        float lim;
        bool has, fits, any;
        int bx0, by0, bx1, by1, w, h;
        for (int attempt = 0;; ++attempt) {
            lim = (U + L) + tc;
            has = pending && (t < lim);  // samples inside this chunk
            bx0 = INT_MAX, by0 = INT_MAX, bx1 = INT_MIN, by1 = INT_MIN;
            if (has) {
                // Samples run over t in [t, min(t_max, pred(lim))]; fp32 o + t*d is monotone in t, so the two end
                // points bound every sample exactly and the texels needed are floor(. - 0.5) of those bounds and
                // their +1 neighbours.
                float tb = fminf(t_max, lim);
                float xa = o0 + t * d0, ya = o1 + t * d1;
                float xb = o0 + tb * d0, yb = o1 + tb * d1;
                float xmin = fminf(xa, xb), xmax = fmaxf(xa, xb);
                float ymin = fminf(ya, yb), ymax = fmaxf(ya, yb);
                if (DERIV) {
                    xmin = fminf(xmin, xmin + d1);
                    xmax = fmaxf(xmax, xmax + d1);
                    ymin = fminf(ymin, ymin - d0);
                    ymax = fmaxf(ymax, ymax - d0);
                }
                bx0 = (int)floorf(xmin - 0.5f);
                bx1 = (int)floorf(xmax - 0.5f) + 1;
                by0 = (int)floorf(ymin - 0.5f);
                by1 = (int)floorf(ymax - 0.5f) + 1;
            }
            bx0 = wave_min_i(bx0);
            by0 = wave_min_i(by0);
            bx1 = wave_max_i(bx1);
            by1 = wave_max_i(by1);
            if (lane == 0) {
                s_box[wave][0] = bx0;
                s_box[wave][1] = by0;
                s_box[wave][2] = bx1;
                s_box[wave][3] = by1;
                s_pend[wave] = pend != 0ull;
            }
            __syncthreads();  // (A) boxes visible; every thread has left the previous chunk's tile
            bx0 = s_box[0][0], by0 = s_box[0][1], bx1 = s_box[0][2], by1 = s_box[0][3];
#pragma unroll
            for (int q = 1; q < RT_WAVES; ++q) {
                bx0 = min(bx0, s_box[q][0]);
                by0 = min(by0, s_box[q][1]);
                bx1 = max(bx1, s_box[q][2]);
                by1 = max(by1, s_box[q][3]);
            }
            any = bx1 >= bx0;
            w = bx1 - bx0 + 1, h = by1 - by0 + 1;
            fits = any && (TRANSP ? (h <= (TILE_S < TILE_W ? TILE_S : TILE_W) && w <= TILE_H)
                                  : (w <= (TILE_S < TILE_W ? TILE_S : TILE_W) && h <= TILE_H));
            if (fits || !any || attempt == 4) break;
            L *= 0.5f;
            __syncthreads();  // everyone has read s_box before it is rewritten
        }
        int any_pending = 0;
#pragma unroll
        for (int q = 0; q < RT_WAVES; ++q) any_pending |= s_pend[q];
        if (!any_pending) break;  // uniform: all lines done
        if (tid == 0) { RSTAT(0, 1); RSTAT(1, fits ? 1 : 0); RSTAT(2, any ? 1 : 0); RSTAT(5, w > 0 ? w : 0); RSTAT(6, h > 0 ? h : 0); }
        if (fits) {
            // Stage the footprint: all of a thread's (up to 36) global loads are issued before the first
            // LDS store so their latencies overlap (a load -> store loop serialises one L2 round trip per
            // element).  Rows by wave-quarter, 32 consecutive texels per half-wave = one 128-B segment.
            // TRANSP: the same with the roles of x and y exchanged, reading the transposed copy of the image (rows of
            // length H), so the loads stay coalesced and the LDS stores conflict-free.
            const int cx = tid & 31, ry = tid >> 5;
            const int fast0 = TRANSP ? by0 : bx0, slow0 = TRANSP ? bx0 : by0;
            const int nfast = TRANSP ? h : w, nslow = TRANSP ? w : h;
            const int Wf = TRANSP ? H : W, Hs = TRANSP ? W : H;
            const float* __restrict__ src = TRANSP ? p.imagesT + (int64_t)blockIdx.z * p.image_stride : img;
            float stage[(TILE_H / RT_STAGE_ROWS) * (TILE_W / 32)];
#pragma unroll
            for (int q = 0; q < TILE_H / RT_STAGE_ROWS; ++q) {
                const int r = ry + RT_STAGE_ROWS * q;
                const int gs = min(max(slow0 + r, 0), Hs - 1);
                const float* __restrict__ row = src + (size_t)gs * Wf;
#pragma unroll
                for (int c3 = 0; c3 < TILE_W / 32; ++c3) {
                    const int c = cx + 32 * c3;
                    const int gf = min(max(fast0 + c, 0), Wf - 1);
                    stage[q * (TILE_W / 32) + c3] = (r < nslow && c < nfast) ? row[gf] : 0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < TILE_H / RT_STAGE_ROWS; ++q) {
                const int r = ry + RT_STAGE_ROWS * q;
#pragma unroll
                for (int c3 = 0; c3 < TILE_W / 32; ++c3) {
                    const int c = cx + 32 * c3;
                    if (r < nslow && c < nfast) tile[r * TILE_S + c] = stage[q * (TILE_W / 32) + c3];
                }
            }
        }
        __syncthreads();  // (B) tile complete; s_box/s_pend may be rewritten by the next chunk
        if (has) {
            if (fits) {
                const float* tile_shifted = tile - (TRANSP ? (bx0 * TILE_S + by0) : (by0 * TILE_S + bx0));
                // t <= t_max && t < lim  <=>  t <= min(t_max, pred(lim)): one compare per step
                const float t_end = fminf(t_max, nextafterf(lim, -FLT_MAX));
                // ref: RadonIntermediate.cu:105-123 (t += step accumulates in fp32)
                for (; t <= t_end; t += RADON_STEP) {
                    RSTAT(4, 1);
                    float x = o0 + t * d0, y = o1 + t * d1;
                    sum += tex_lds<TILE_S, TRANSP>(tile_shifted, x, y);
                    if (DERIV) sumo += tex_lds<TILE_S, TRANSP>(tile_shifted, x + d1, y - d0);
                }
            } else {
                for (; t <= t_max && t < lim; t += RADON_STEP) {
                    RSTAT(3, 1);
                    float x = o0 + t * d0, y = o1 + t * d1;
                    sum += tex_global(img, W, H, x, y);
                    if (DERIV) sumo += tex_global(img, W, H, x + d1, y - d0);
                }
            }
        }
        U += L;
    }
    // Safety net (never taken for sane sizes): finish whatever MAX_CHUNKS did not cover.
    if (active)
        for (; t <= t_max; t += RADON_STEP) {
            float x = o0 + t * d0, y = o1 + t * d1;
            sum += tex_global(img, W, H, x, y);
            if (DERIV) sumo += tex_global(img, W, H, x + d1, y - d0);
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
    // line normal of this workgroup's first angle: (nx, ny) = (-sin a, cos a)
    const int ix0 = min((int)blockIdx.x * RT_A, p.n_alpha - 1);
    const float nx = -p.trig[2 * ix0], ny = p.trig[2 * ix0 + 1];
    // Row stride.  A half-wave is two "combs" (two adjacent angles) of 16 points spaced 1.9 px along the line normal; the
    // two combs nearly coincide, so any bank function gives at least a 2-way conflict between them (simulation:
    // scripts/analysis/radon_lds_layouts.py).  With stride 96 the bank is the texel's index along the tile's fast axis
    // alone: a comb whose normal is closer to that axis than to the other advances >= 1.33 banks per lane over < 32 banks,
    // conflict-free in itself, and the pass costs exactly two cycles.  So: normal closer to x -> plain tile, stride 96;
    // normal closer to y -> the tile is staged TRANSPOSED (from the transposed copy of the image stack made by
    // ecc_launch_radon), stride 96 again.  Without the transposed copy (p.imagesT null) those workgroups keep the
    // 97 / 95 rule (the bank advances along x+y or x-y, whichever the normal is closer to).
    // Measured per 1024^2 image: 97/95 rule everywhere 0.771 ms, stride 96 for x-normals only 0.740 ms, with the transposed
    // tile for y-normals as well 0.699 ms.
    // (On top of this a texel-pair tile -- element = float2 {T(c), T(c+1)} along the fast axis, one aligned ds_read_b64 per
    // footprint row at 256 B/clk instead of ds_read2_b32's 128 -- was measured at the same LDS budget, 72 slow x 64 fast
    // elements = 36.9 KB, 4 workgroups per CU: bit-exact and 1.010 ms.  The chunks get 0.55-0.74x as long and the cost per
    // chunk -- bounding box, staging, three barriers -- outweighs the halved read cycles; round 1 had found the same for
    // the full-size pair tile at 2 workgroups per CU, 1.12 ms.  The chunk machinery alone -- the kernel with its sampling
    // loop skipped -- takes 0.103 ms per image: 15 % of the kernel, and about what separates it from its LDS time.)
    if (fabsf(nx) >= fabsf(ny))
        radon_body<DERIV, TILE_W>(p, sh);
    else if (p.imagesT)
        radon_body<DERIV, TILE_W, true>(p, sh);
    else if (fabsf(nx + ny) >= fabsf(nx - ny))
        radon_body<DERIV, TILE_W + 1>(p, sh);
    else
        radon_body<DERIV, TILE_W - 1>(p, sh);
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
