// preprocess_kernel.hip -- projection pre-processing on the device, the step in front of the Radon
// intermediate.
//
// The reference does this on the host, one image at a time, in five sweeps over the image plus two
// convolution passes (ref: LibEpipolarConsistency/Gui/PreProccess.cpp:57-166, HeaderOnly/NRRD/nrrd_lowpass.hxx:45-79)
// and then uploads the result for the Radon kernel (RadonIntermediate.cpp:27).  Here the whole chain is ONE
// kernel per batch, HBM-bound by construction (4 B read + 4 B written per pixel, halo re-reads come out of L2):
//   workgroup = 64 x 64 output pixels; it stages the (64+2k)^2 input footprint into LDS *through* the pixel-wise
//   stages (intensity -> border zero/feather -> blanks, evaluated at the un-flipped source position, i.e. the
//   flips are an index map), runs the horizontal Gaussian pass LDS -> LDS, the vertical pass LDS -> registers,
//   multiplies with the cosine weight and writes each pixel once.
// Arithmetic is the reference's, expression by expression (float pixel ops; the two convolution sums in binary64
// over o = -k .. k-1, the dropped last tap included; clamp addressing), so results are bit-identical to
// oracle/ecc_oracle.c (eccor_preprocess + eccor_cos_weight).
#include <hip/hip_runtime.h>
#include <math.h>

#include "ecc_layout.h"

namespace {

constexpr int PP_TW = 64, PP_TH = 64, PP_THREADS = 256;  // PP_TW = one wave per tile row

// ref: Gui/PreProccess.cpp:8-13 (weighting, double)
__device__ __forceinline__ double weighting_d(double x)
{
    if (x < -1.0 || x > 1.0) return 0;
    const double xx = x * x;
    return 1.0 - 2 * xx + xx * xx;
}

// Pixel-wise part of PreProccess::process for the pixel at SOURCE position (sx, sy)
// (ref: Gui/PreProccess.cpp:78-120, same order: intensity, left, right, bottom, top, blanks).
__device__ __forceinline__ float pointwise(const EccPreprocessParams& p, float v, int sx, int sy, float scale, float bias)
{
    float pixel = v * scale + bias;
    if (p.apply_log) pixel = (float)-(float)log((double)pixel);
    if (pixel < 0 || isnan(pixel) || isinf(pixel)) pixel = 0;
    if (sx < p.zero[0] + p.feather[0])
        pixel *= sx <= p.zero[0] ? 0 : (float)weighting_d(1 - (float)(sx - p.zero[0]) / p.feather[0]);
    {
        const int b = p.n_u - sx;
        if (b <= p.zero[1] + p.feather[1])
            pixel *= b <= p.zero[1] ? 0 : (float)weighting_d(1 - (float)(b - p.zero[1]) / p.feather[1]);
    }
    {
        const int b = p.n_v - sy;
        if (b <= p.zero[2] + p.feather[2])
            pixel *= b <= p.zero[2] ? 0 : (float)weighting_d(1 - (float)(b - p.zero[2]) / p.feather[2]);
    }
    if (sy < p.zero[3] + p.feather[3])
        pixel *= sy <= p.zero[3] ? 0 : (float)weighting_d(1 - (float)(sy - p.zero[3]) / p.feather[3]);
    for (int q = 0; q < p.n_blanks; ++q) {
        const int* bl = p.blanks + 4 * q;
        // ref: :116-119 (the y loop is bounded by img.size(0) there; clipped to the image as well)
        if (sx >= bl[0] && sx < bl[2] && sy >= bl[1] && sy < bl[3] && sy < p.n_u) pixel = 0;
    }
    return pixel;
}

// KT: half kernel width known at compile time (taps live in registers, loops unroll), or -1 = runtime k with
// the taps read from an LDS copy (a scalar load per tap inside the loop serialises on its latency: 10.8 -> x us).
template <int KT>
__global__ __launch_bounds__(PP_THREADS) void preprocess_kernel(EccPreprocessParams p)
{
    extern __shared__ float lds[];
    const int k = KT >= 0 ? KT : p.k;
    const int AW = PP_TW + 2 * k, AH = PP_TH + 2 * k;
    float* A = lds;            // AH x AW : pixel-wise result incl. halo (clamped = the convolution's clamp addressing)
    float* B = lds + AH * AW;  // AH x PP_TW : after the horizontal pass
    double* taps_lds = reinterpret_cast<double*>(B + AH * PP_TW);  // runtime-k path only (8-byte aligned: even counts)
    double taps[KT > 0 ? 2 * KT : 1];
    if (KT > 0) {
#pragma unroll
        for (int o = 0; o < 2 * KT; ++o) taps[o] = p.kernel[o];
    } else if (KT < 0) {
        if ((int)threadIdx.x < 2 * k) taps_lds[threadIdx.x] = p.kernel[threadIdx.x];
    }
    const int img_i = blockIdx.z;
    const float* __restrict__ src = p.in + (int64_t)img_i * p.stride;
    float* __restrict__ dst = p.out + (int64_t)img_i * p.stride;
    const int x0 = blockIdx.x * PP_TW, y0 = blockIdx.y * PP_TH;
    const int W = p.n_u, H = p.n_v;

    float scale = p.scale, bias = p.bias;
    if (p.normalize) {  // ref: :68-76
        bias = 0;
        scale = p.scale / p.max_d[img_i];
    }

    // thread (tx, ty) walks rows ty, ty+4, ... and columns tx, tx+64, ...: no integer divisions, and the 64
    // lanes of a wave read 64 consecutive texels (reversed when flipped)
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (KT > 0) {
        // compile-time footprint: ALL of a thread's loads are issued before the first is consumed (a load -> pointwise ->
        // store loop waits one memory round trip per element, ~20 per thread): 9.8 -> 7.9 us per 1024^2 image with the
        // default low-pass.  (Without a low-pass the plain loop below is faster, 3.2 against 6.1 us.)
        constexpr int AHc = PP_TH + 2 * (KT > 0 ? KT : 0), AWc = PP_TW + 2 * (KT > 0 ? KT : 0);
        constexpr int NR = (AHc + 3) / 4, NC = (AWc + 63) / 64;
        float v[NR][NC];
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int ly = ty + 4 * q;
            const int gy = min(max(y0 + ly - k, 0), H - 1);
            const int sy = p.flip_v ? H - 1 - gy : gy;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int lx = tx + 64 * c;
                const int gx = min(max(x0 + lx - k, 0), W - 1);
                const int sx = p.flip_u ? W - 1 - gx : gx;
                v[q][c] = (ly < AHc && lx < AWc) ? src[(size_t)sy * W + sx] : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int ly = ty + 4 * q;
            const int gy = min(max(y0 + ly - k, 0), H - 1);
            const int sy = p.flip_v ? H - 1 - gy : gy;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int lx = tx + 64 * c;
                if (ly < AHc && lx < AWc) {
                    const int gx = min(max(x0 + lx - k, 0), W - 1);
                    const int sx = p.flip_u ? W - 1 - gx : gx;
                    float val = v[q][c];
                    if (p.process) val = pointwise(p, val, sx, sy, scale, bias);
                    A[ly * AW + lx] = val;
                }
            }
        }
    } else {
        for (int ly = ty; ly < AH; ly += PP_THREADS / 64) {
            const int gy = min(max(y0 + ly - k, 0), H - 1);
            const int sy = p.flip_v ? H - 1 - gy : gy;  // ref: :123-136
            for (int lx = tx; lx < AW; lx += 64) {
                const int gx = min(max(x0 + lx - k, 0), W - 1);
                const int sx = p.flip_u ? W - 1 - gx : gx;
                float v = src[(size_t)sy * W + sx];
                if (p.process) v = pointwise(p, v, sx, sy, scale, bias);
                A[ly * AW + lx] = v;
            }
        }
    }
    __syncthreads();
    if (k > 0) {
        // horizontal pass, ref: nrrd_lowpass.hxx:52-63 (o = -kx .. kx-1, double sum, result cast to T)
        for (int ly = ty; ly < AH; ly += PP_THREADS / 64) {
            double sum = 0;
            if (KT > 0) {
#pragma unroll
                for (int o = 0; o < 2 * KT; ++o) sum += A[ly * AW + tx + o] * taps[o];
            } else {
                for (int o = 0; o < 2 * k; ++o) sum += A[ly * AW + tx + o] * taps_lds[o];
            }
            B[ly * PP_TW + tx] = (float)sum;
        }
        __syncthreads();
    }
    const float sdd = p.cosw ? p.cosw[3 * img_i] : 0.f;
    const float ppu = p.cosw ? p.cosw[3 * img_i + 1] : 0.f, ppv = p.cosw ? p.cosw[3 * img_i + 2] : 0.f;
    const bool weight = p.cosw && p.cosw_valid[img_i];
    for (int ly = ty; ly < PP_TH; ly += PP_THREADS / 64) {
        const int lx = tx;
        const int gx = x0 + lx, gy = y0 + ly;
        if (gx >= W || gy >= H) continue;
        float pixel;
        if (k > 0) {
            // vertical pass, ref: nrrd_lowpass.hxx:64-75 (uses kernelx again, o = -ky .. ky-1)
            double sum = 0;
            if (KT > 0) {
#pragma unroll
                for (int o = 0; o < 2 * KT; ++o) sum += B[(ly + o) * PP_TW + lx] * taps[o];
            } else {
                for (int o = 0; o < 2 * k; ++o) sum += B[(ly + o) * PP_TW + lx] * taps_lds[o];
            }
            pixel = (float)sum;
        } else {
            pixel = A[ly * AW + lx];
        }
        if (weight) {  // ref: Gui/PreProccess.cpp:156-165
            const float pou = (float)gx - ppu;
            const float pov = (float)gy - ppv;
            const float cos_weight = sdd / sqrtf(pou * pou + pov * pov + sdd * sdd);
            pixel *= cos_weight;
        }
        dst[(size_t)gy * W + gx] = pixel;
    }
}

// ref: Gui/PreProccess.cpp:68-71: max = img[0]; if (img[i] > max) max = img[i]  (NaNs never win; a NaN first
// pixel stays).  One workgroup per image.
__global__ __launch_bounds__(1024) void image_max_kernel(const float* __restrict__ in, int64_t stride, int64_t len,
                                                         float* __restrict__ max_out)
{
    __shared__ float s[1024 / 64];
    const float* img = in + (int64_t)blockIdx.x * stride;
    float m = -INFINITY;
    for (int64_t i = threadIdx.x; i < len; i += 1024) {
        const float v = img[i];
        if (v > m) m = v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(m, off);
        if (o > m) m = o;
    }
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 1024 / 64; ++w)
            if (s[w] > m) m = s[w];
        const float first = img[0];
        max_out[blockIdx.x] = isnan(first) ? first : m;
    }
}

}  // namespace

extern "C" size_t ecc_preprocess_lds_bytes(int k)
{
    const int AW = PP_TW + 2 * k, AH = PP_TH + 2 * k;
    return sizeof(float) * ((size_t)AH * AW + (k > 0 ? (size_t)AH * PP_TW : 0)) + sizeof(double) * 2 * (size_t)k;
}

extern "C" hipError_t ecc_launch_preprocess(const EccPreprocessParams* p, hipStream_t stream)
{
    if (p->normalize && p->process) {
        hipLaunchKernelGGL(image_max_kernel, dim3(p->n_img), dim3(1024), 0, stream, p->in, p->stride,
                           (int64_t)p->n_u * p->n_v, p->max_d);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    dim3 grid((p->n_u + PP_TW - 1) / PP_TW, (p->n_v + PP_TH - 1) / PP_TH, p->n_img);
    const size_t lds = ecc_preprocess_lds_bytes(p->k);
    if (p->k == 0)
        hipLaunchKernelGGL(preprocess_kernel<0>, grid, dim3(PP_THREADS), lds, stream, *p);
    else if (p->k == 5)  // the reference's default (Gui/PreProccess.h:28)
        hipLaunchKernelGGL(preprocess_kernel<5>, grid, dim3(PP_THREADS), lds, stream, *p);
    else
        hipLaunchKernelGGL(preprocess_kernel<-1>, grid, dim3(PP_THREADS), lds, stream, *p);
    return hipGetLastError();
}
