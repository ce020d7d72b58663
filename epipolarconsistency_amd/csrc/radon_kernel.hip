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
//     an LDS tile (4 taps = 2 x ds_read2_b32) instead of 4 scattered global loads;
//   * a workgroup owns RT_A adjacent angles x RT_T adjacent distances (256 threads).  Its lines are
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

namespace {

constexpr int RT_T = 32;                   // distance bins per workgroup (lane & 31)
constexpr int RT_A = 8;                    // angle bins per workgroup    (tid >> 5)
constexpr int RT_THREADS = RT_T * RT_A;    // 256
constexpr int TILE_S = 96;                 // LDS tile row stride (floats)
constexpr int TILE_H = 96;                 // LDS tile rows
constexpr float RADON_STEP = .66f;         // ref: RadonIntermediate.cu:102
constexpr int MAX_CHUNKS = 8192;           // bound on the chunk loop (every spin is bounded)

// Exact fp32 bilinear rule on global memory with clamp addressing; slow path used
// only when a chunk's footprint does not fit the LDS tile.
__device__ __forceinline__ float tex_global(const float* __restrict__ img, int W, int H, float x, float y)
{
    float xb = x - 0.5f, yb = y - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    float fx = xb - fi, fy = yb - fj;
    int i = (int)fi, j = (int)fj;
    int i0 = min(max(i, 0), W - 1), i1 = min(max(i + 1, 0), W - 1);
    int j0 = min(max(j, 0), H - 1), j1 = min(max(j + 1, 0), H - 1);
    float T00 = img[(size_t)j0 * W + i0], T10 = img[(size_t)j0 * W + i1];
    float T01 = img[(size_t)j1 * W + i0], T11 = img[(size_t)j1 * W + i1];
    float r0 = (1.f - fx) * T00 + fx * T10;
    float r1 = (1.f - fx) * T01 + fx * T11;
    return (1.f - fy) * r0 + fy * r1;
}

// Same rule on the staged tile.  tile_off = by0*TILE_S + bx0 (tile origin in image texels); the
// tile already holds clamped (replicated) texels, so taps need no index clamps.
__device__ __forceinline__ float tex_lds(const float* tile, int tile_off, float x, float y)
{
    float xb = x - 0.5f, yb = y - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    float fx = xb - fi, fy = yb - fj;
    // fj*TILE_S + fi is an exact small integer in fp32 (|.| < 2^24) whether or not it is fused.
    int idx = (int)(fj * (float)TILE_S + fi) - tile_off;
    float T00 = tile[idx], T10 = tile[idx + 1];
    float T01 = tile[idx + TILE_S], T11 = tile[idx + TILE_S + 1];
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

template <bool DERIV>
__global__ __launch_bounds__(RT_THREADS) void radon_kernel(EccRadonParams p)
{
    __shared__ float tile[TILE_S * TILE_H];
    __shared__ int s_box[4][4];      // per wave: min x, min y, max x, max y
    __shared__ float s_u[4][2];      // per wave: min/max of the along-line coordinate
    __shared__ float s_L;
    __shared__ int s_pend[4];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int ix = blockIdx.x * RT_A + (tid >> 5);
    const int iy = blockIdx.y * RT_T + (tid & 31);
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
    const float U0 = fminf(fminf(s_u[0][0], s_u[1][0]), fminf(s_u[2][0], s_u[3][0]));
    const float U1 = fmaxf(fmaxf(s_u[0][1], s_u[1][1]), fmaxf(s_u[2][1], s_u[3][1]));
    if (tid == 0) {
        // Chunk length so that the bounding box of a (L x band) rectangle at this angle fits the tile.
        const float Pi = 3.14159265359f;
        float dtau = diag / (float)p.n_t;
        float far = fmaxf(fabsf(U0), fabsf(U1));
        float band = (RT_T - 1) * dtau + 2.f + (RT_A - 1) * (Pi / (float)p.n_alpha) * far;
        float cs = fabsf(d0), sn = fabsf(d1);
        float Lw = cs > 1e-3f ? ((float)(TILE_S - 4) - band * sn) / cs : 1e9f;
        float Lh = sn > 1e-3f ? ((float)(TILE_H - 4) - band * cs) / sn : 1e9f;
        s_L = fminf(fmaxf(fminf(Lw, Lh), 4.f), 4096.f);
    }
    __syncthreads();
    const float L = s_L;

    float sum = 0.f, sumo = 0.f;
    float U = U0;
    for (int it = 0; it < MAX_CHUNKS; ++it, U += L) {
        const float lim = (U + L) + tc;
        const bool pending = active && (t <= t_max);  // samples left at all
        const bool has = pending && (t < lim);        // samples inside this chunk
        int bx0 = INT_MAX, by0 = INT_MAX, bx1 = INT_MIN, by1 = INT_MIN;
        if (has) {
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
            bx0 = (int)floorf(xmin - 0.5f) - 1;
            bx1 = (int)floorf(xmax - 0.5f) + 2;
            by0 = (int)floorf(ymin - 0.5f) - 1;
            by1 = (int)floorf(ymax - 0.5f) + 2;
        }
        bx0 = wave_min_i(bx0);
        by0 = wave_min_i(by0);
        bx1 = wave_max_i(bx1);
        by1 = wave_max_i(by1);
        const unsigned long long pend = __ballot(pending);
        if (lane == 0) {
            s_box[wave][0] = bx0;
            s_box[wave][1] = by0;
            s_box[wave][2] = bx1;
            s_box[wave][3] = by1;
            s_pend[wave] = pend != 0ull;
        }
        __syncthreads();  // (A) boxes visible; every thread has left the previous chunk's tile
        if (!(s_pend[0] | s_pend[1] | s_pend[2] | s_pend[3])) break;  // uniform: all lines done
        bx0 = min(min(s_box[0][0], s_box[1][0]), min(s_box[2][0], s_box[3][0]));
        by0 = min(min(s_box[0][1], s_box[1][1]), min(s_box[2][1], s_box[3][1]));
        bx1 = max(max(s_box[0][2], s_box[1][2]), max(s_box[2][2], s_box[3][2]));
        by1 = max(max(s_box[0][3], s_box[1][3]), max(s_box[2][3], s_box[3][3]));
        const bool any = bx1 >= bx0;
        const int w = bx1 - bx0 + 1, h = by1 - by0 + 1;
        const bool fits = any && w <= TILE_S && h <= TILE_H;
        if (fits) {
            const int cx = tid & 31, ry = tid >> 5;
            for (int r = ry; r < h; r += RT_THREADS / 32) {
                int gy = min(max(by0 + r, 0), H - 1);
                const float* __restrict__ row = img + (size_t)gy * W;
                for (int c = cx; c < w; c += 32) {
                    int gx = min(max(bx0 + c, 0), W - 1);
                    tile[r * TILE_S + c] = row[gx];
                }
            }
        }
        __syncthreads();  // (B) tile complete; s_box/s_pend may be rewritten by the next chunk
        if (has) {
            if (fits) {
                const int tile_off = by0 * TILE_S + bx0;
                // ref: RadonIntermediate.cu:105-123 (t += step accumulates in fp32)
                for (; t <= t_max && t < lim; t += RADON_STEP) {
                    float x = o0 + t * d0, y = o1 + t * d1;
                    sum += tex_lds(tile, tile_off, x, y);
                    if (DERIV) sumo += tex_lds(tile, tile_off, x + d1, y - d0);
                }
            } else {
                for (; t <= t_max && t < lim; t += RADON_STEP) {
                    float x = o0 + t * d0, y = o1 + t * d1;
                    sum += tex_global(img, W, H, x, y);
                    if (DERIV) sumo += tex_global(img, W, H, x + d1, y - d0);
                }
            }
        }
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
            else if (p.post_process == 2) result = result < 0 ? -logf(-result + 1) : logf(result + 1);
        }
        float* out = p.out + (int64_t)blockIdx.z * p.out_stride;
        out[(size_t)(ix + 1) * p.pitch + (iy + 1)] = result;
    }
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
