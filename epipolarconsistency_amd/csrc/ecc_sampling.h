// ecc_sampling.h -- the normative bilinear rule on a plain row-major image in global memory.
//
// Replaces CUDA's un-normalised, clamped, linearly filtered texture fetch (ref:
// LibUtilsCuda/CudaBindlessTexture.cpp:25-39) by the exact fp32 rule of SURVEY.md 8c:
// xb = x - .5, yb = y - .5; taps at the clamped floor / floor + 1; weights are the exact fractions.
#ifndef ECC_SAMPLING_H
#define ECC_SAMPLING_H

#include <hip/hip_runtime.h>

// The two arithmetic conventions of the bilinear rule (ecc_radon_set_arithmetic):
//   exact (FMA = false): (1 - fx) * T00 + fx * T10 ..., every product and sum rounded -- the CPU reading of the source,
//                        bit for bit the oracle's eccor_tex2d;
//   contracted (FMA = true): T00 + fx * (T10 - T00) as one fused multiply-add per lerp, three differences and three
//                        fmaf -- bit for bit the oracle's or_tex2d_contract.
template <bool FMA>
__device__ __forceinline__ float ecc_bilerp(float fx, float fy, float T00, float T10, float T01, float T11)
{
    if (FMA) {
        const float r0 = __builtin_fmaf(fx, T10 - T00, T00);
        const float r1 = __builtin_fmaf(fx, T11 - T01, T01);
        return __builtin_fmaf(fy, r1 - r0, r0);
    }
    const float r0 = (1.f - fx) * T00 + fx * T10;
    const float r1 = (1.f - fx) * T01 + fx * T11;
    return (1.f - fy) * r0 + fy * r1;
}

template <bool FMA = false>
__device__ __forceinline__ float ecc_tex_global(const float* __restrict__ img, int W, int H, float x, float y)
{
    float xb = x - 0.5f, yb = y - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    float fx = xb - fi, fy = yb - fj;
    // float -> int saturates on the device; the clamps do the rest
    int i = (int)fi, j = (int)fj;
    int i0 = min(max(i, 0), W - 1), i1 = min(max(i + 1, 0), W - 1);
    int j0 = min(max(j, 0), H - 1), j1 = min(max(j + 1, 0), H - 1);
    float T00 = img[(size_t)j0 * W + i0], T10 = img[(size_t)j0 * W + i1];
    float T01 = img[(size_t)j1 * W + i0], T11 = img[(size_t)j1 * W + i1];
    return ecc_bilerp<FMA>(fx, fy, T00, T10, T01, T11);
}

// Same rule with explicit element strides: texel (i, j) lives at img[i * si + j * sj].  (si, sj) = (1, W) is the
// row-major image, (H, 1) its transposed copy -- MetricDirect picks per line whichever makes the wave's lanes
// (adjacent, nearly parallel lines at the same step) land in the same cache lines.  Identical arithmetic.
__device__ __forceinline__ float ecc_tex_global_strided(const float* __restrict__ img, int W, int H, int si, int sj, float x,
                                                        float y)
{
    float xb = x - 0.5f, yb = y - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    float fx = xb - fi, fy = yb - fj;
    int i = (int)fi, j = (int)fj;
    int i0 = min(max(i, 0), W - 1), i1 = min(max(i + 1, 0), W - 1);
    int j0 = min(max(j, 0), H - 1), j1 = min(max(j + 1, 0), H - 1);
    float T00 = img[(size_t)j0 * sj + (size_t)i0 * si], T10 = img[(size_t)j0 * sj + (size_t)i1 * si];
    float T01 = img[(size_t)j1 * sj + (size_t)i0 * si], T11 = img[(size_t)j1 * sj + (size_t)i1 * si];
    float r0 = (1.f - fx) * T00 + fx * T10;
    float r1 = (1.f - fx) * T01 + fx * T11;
    return (1.f - fy) * r0 + fy * r1;
}

#endif
