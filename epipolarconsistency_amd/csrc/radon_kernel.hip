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
// fp32 rule of SURVEY.md 8c, evaluated from LDS): a workgroup owns 16 adjacent angles x 16 adjacent distances
// (256 threads); its lines are nearly parallel and sweep a narrow band of the image, which is walked slab by slab
// through an LDS tile of texel pairs (ecc_slab_tile.h: ds_read_b64 per footprint row, linear tile with sliding row
// windows, analytic slab geometry with a per-thread containment check, register prefetch of the next slab).  The
// kernel is bound by vector-ALU issue, not by the LDS pipe.
//   * sin/cos of the bin angle come from a host table (one entry per angle), not per thread;
//   * output goes to the transposed, border-padded layout of ecc_layout.h (distance fastest).
#include <hip/hip_runtime.h>
#include <float.h>
#include <limits.h>

#include "ecc_layout.h"

#ifdef ECC_RADON_STATS
// 0: slabs, 1: LDS-path steps, 2: global-path steps inside slabs, 3: steps of the safety net, 4: sum of S, 5: sum of H
__device__ unsigned long long g_radon_stats[8];
#define ECC_SLAB_STAT(i, v) atomicAdd(&g_radon_stats[i], (unsigned long long)(v))
#endif
#include "ecc_slab_tile.h"

namespace {

using namespace ecc_slab;

constexpr int RT_T = 16;                   // distance bins per workgroup
constexpr int RT_A = 16;                   // angle bins per workgroup
constexpr int RT_THREADS = RT_T * RT_A;    // 256
static_assert(RT_THREADS == ecc_slab::THREADS, "one line per thread of the slab walker");
constexpr float RADON_STEP = .66f;         // ref: RadonIntermediate.cu:102

// The workgroup's band from its two corner angles and two corner distances (all values identical in every thread).
// A sample of the line with normal (l0, l1) and offset c (position = c * normal + (.5, .5) + t * direction) at slow
// coordinate s has fast coordinate f = .5 + (s - .5) * m + c / lf, with lf the normal's f-component and m = -ls / lf.
struct CornerBand {
    float beta;
    float mk[2];           // m - beta of the first / last angle
    float g0[2];           // .5 - .5 m
    float cl[2], ch[2];    // c / lf at the workgroup's smallest / largest offset (either order)
    __device__ __forceinline__ void operator()(float sa, float sb, float& gmin, float& gmax) const
    {
        gmin = FLT_MAX;
        gmax = -FLT_MAX;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float u0 = mk[e] * sa, u1 = mk[e] * sb;
            gmin = fminf(gmin, g0[e] + fminf(u0, u1) + fminf(cl[e], ch[e]));
            gmax = fmaxf(gmax, g0[e] + fmaxf(u0, u1) + fmaxf(cl[e], ch[e]));
        }
    }
};

struct RadonSums {
    float sum = 0.f, sumo = 0.f;
    // ref: RadonIntermediate.cu:105-123
    __device__ __forceinline__ void add(float vA, float vB, float) { sum += vA; sumo += vB; }
};

template <bool DERIV, bool TRANSP, bool FMA>
__device__ __forceinline__ void radon_body(const EccRadonParams& p, Shared& sh)
{
    const int tid = threadIdx.x;
    // a 32-lane group (what an LDS read is serviced in) is 16 adjacent distances x 2 adjacent angles
    const int ix = blockIdx.x * RT_A + ((tid >> 4) & (RT_A - 1));
    const int iy = blockIdx.y * RT_T + (tid & 15);
    const float* __restrict__ img = p.images + (int64_t)blockIdx.z * p.image_stride;
    const int W = p.n_u, H = p.n_v;
    const float n_u = (float)W, n_v = (float)H;
    const bool in_range = ix < p.n_alpha && iy < p.n_t;

    // ---- per-bin line set-up, ref: RadonIntermediate.cu:36-99 (same expressions, same order) ----
    LineRun ln = {0.f, 0.f, 1.f, 0.f, 0.f, -1.f, false};
    const float diag = sqrtf(n_u * n_u + n_v * n_v);
    if (in_range) {
        float y_rel = (iy / (float)p.n_t - 0.5f);
        float tau = y_rel * diag;
        float l0 = -p.trig[2 * ix];
        float l1 = p.trig[2 * ix + 1];
        float l2 = -tau;
        l2 += -0.5f * n_u * l0 - 0.5f * n_v * l1;
        float o0 = -l2 * l0;
        float o1 = -l2 * l1;
        const float d0 = l1;
        const float d1 = -l0;
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
        const float t = ts[1], t_max = ts[2];
        float u = o0 + t * d0, v = o1 + t * d1;
        bool inb = (u <= n_u && v <= n_v && u >= 0 && v >= 0);
        o0 += .5f;
        o1 += .5f;
        if (DERIV) {
            o0 -= .5f * d1;
            o1 += .5f * d0;
        }
        ln = {o0, o1, d0, d1, t, t_max, inb && !(t_max <= t)};
    }

    // ---- the workgroup's band (identical in every thread) ----
    const float* __restrict__ src = TRANSP ? (p.imagesT + (int64_t)blockIdx.z * p.image_stride) : img;
    const int ixA = blockIdx.x * RT_A, ixB = min(ixA + RT_A - 1, p.n_alpha - 1), ixM = (ixA + ixB) >> 1;
    const int iyA = blockIdx.y * RT_T, iyB = min(iyA + RT_T - 1, p.n_t - 1);
    BandFrame bf;
    CornerBand bd;
    {
        const float lfM = TRANSP ? p.trig[2 * ixM + 1] : -p.trig[2 * ixM];
        const float lsM = TRANSP ? -p.trig[2 * ixM] : p.trig[2 * ixM + 1];
        // direction (d0, d1) = (l1, -l0): its s-component is -l0 (plain) or l1 (transposed)
        const float dsM = TRANSP ? lfM : -lfM;
        bf.sigma = dsM >= 0.f ? 1.f : -1.f;
        set_beta(bf, -lsM / lfM);
        bd.beta = bf.beta;
        const float tauA = ((float)iyA / (float)p.n_t - 0.5f) * diag, tauB = ((float)iyB / (float)p.n_t - 0.5f) * diag;
        const float Pi = 3.14159265359f;
        const float da = (float)(ixB - ixA) * (Pi / (float)p.n_alpha);
        bf.marg = uni(.75f + .6f * diag * da * da);  // the angles between the two corner angles, rounding
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int ixe = e ? ixB : ixA;
            const float l0 = -p.trig[2 * ixe], l1 = p.trig[2 * ixe + 1];
            const float lf = TRANSP ? l1 : l0, ls = TRANSP ? l0 : l1;
            const float inv = 1.f / lf, m = -ls * inv;
            const float kc = 0.5f * n_u * l0 + 0.5f * n_v * l1;
            bd.mk[e] = uni(m - bf.beta);
            bd.g0[e] = uni(.5f - .5f * m);
            bd.cl[e] = uni((tauA + kc - .5f) * inv);
            bd.ch[e] = uni((tauB + kc + .5f) * inv);
        }
    }

    RadonSums acc;
    // second sample of the derivative pair: (x + d1, y - d0)
    walk<TRANSP, DERIV, false, FMA>(sh, img, W, H, src, bf, bd, ln, 0.f, 0.f, ln.d1, -ln.d0, RADON_STEP, acc);

    if (in_range) {
        float result;
        if (!ln.active) result = 0.f;
        else if (!DERIV) result = acc.sum * RADON_STEP;
        else {
            result = (acc.sum - acc.sumo) * RADON_STEP;  // ref: RadonIntermediate.cu:125-140
            if (p.post_process == 1) result = result < 0 ? -sqrtf(-result) : sqrtf(result);
            // logarithm correctly rounded (binary64, rounded once) like the oracle's: once per bin
            else if (p.post_process == 2) result = result < 0 ? -(float)log((double)(-result + 1)) : (float)log((double)(result + 1));
        }
        float* out = p.out + (int64_t)blockIdx.z * p.out_stride;
        out[(size_t)(ix + 1) * p.pitch + (iy + 1)] = result;
    }
}

// FMA: the sampling loop in contracted arithmetic (ecc_radon_set_arithmetic(ECC_RADON_FMA); oracle:
// eccor_set_radon_contract(1)) -- positions fmaf(t, d, o), lerps T00 + fx * (T10 - T00) as one fma each: 40 instead of
// 52 vector instructions per step.  The per-bin set-up above the loop is the same unfused code in both.
#ifndef ECC_RADON_MIN_WAVES
#define ECC_RADON_MIN_WAVES 1
#endif
template <bool DERIV, bool FMA>
__global__ __launch_bounds__(RT_THREADS, ECC_RADON_MIN_WAVES) void radon_kernel(EccRadonParams p)
{
    __shared__ Shared sh;
    // Line normal of the workgroup's middle angle: (nx, ny) = (-sin a, cos a).  The tile's fast axis is the image
    // axis the normal is closer to: a 32-lane group is 16 adjacent distances x 2 adjacent angles, i.e. two nearly
    // coincident "combs" of 16 points spaced 1.9 px along the normal; with the bank = fast index mod 32 a comb
    // advances >= 1.33 banks per lane over < 32 banks, conflict-free in itself, and the two combs cost exactly two
    // passes (scripts/analysis/radon_lds_layouts.py: no bank function of (x, y) separates them).  Normals closer
    // to y read a transposed copy of the image stack (made by ecc_launch_radon's caller), so staging stays coalesced.
    const int ixA = blockIdx.x * RT_A, ixB = min(ixA + RT_A - 1, p.n_alpha - 1), ixM = (ixA + ixB) >> 1;
    const float nx = -p.trig[2 * ixM], ny = p.trig[2 * ixM + 1];
    if (fabsf(nx) >= fabsf(ny))
        radon_body<DERIV, false, FMA>(p, sh);
    else
        radon_body<DERIV, true, FMA>(p, sh);
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
    if (p->arithmetic == 1) {
        if (derivative)
            hipLaunchKernelGGL((radon_kernel<true, true>), grid, block, 0, stream, *p);
        else
            hipLaunchKernelGGL((radon_kernel<false, true>), grid, block, 0, stream, *p);
    } else {
        if (derivative)
            hipLaunchKernelGGL((radon_kernel<true, false>), grid, block, 0, stream, *p);
        else
            hipLaunchKernelGGL((radon_kernel<false, false>), grid, block, 0, stream, *p);
    }
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
