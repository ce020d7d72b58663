// ramp_kernel.hip -- Filter::Ramp: ramp filter along the distance axis of a Radon transform, gfx950.
//
// The reference does it with two batched cuFFT plans and a pointwise kernel between them
// (ref: LibEpipolarConsistency/RadonIntermediate.cu:173-237, apply1DRampFilter): per angle column an
// unnormalised R2C FFT over the n_t distances, bin k = 0..n_t/2 multiplied by the float factor
// k * (-0.5f/(n_t*(n_t/2+1))), unnormalised C2R FFT back.  That is one fixed linear map per column -- a
// CIRCULAR CONVOLUTION with the real, even kernel
//     h[m] = sum_{k=0}^{n_t-1} w_k cos(2 pi k m / n_t),  w_k = (float)min(k, n_t-k) * scale,
// and at n_t = 768 the direct form costs 768 multiply-adds per output: 0.45 GFMA per dtr, a few tens of
// microseconds of vector ALU next to the ~0.9 ms Radon kernel.  So there is no FFT library here:
//   * in the private slab layout (ecc_layout.h) distance is the FAST axis, so one angle row is one
//     contiguous vector: a workgroup owns a row, stages it in LDS (x[s] is then a broadcast read),
//     and every thread produces outputs t, t+256, ... of that row;
//   * h comes from the host as a doubled table h2[m] = h[m mod n_t], m < 2 n_t, in binary64, so
//     h2[t - s + n_t] needs no modulo and consecutive lanes read consecutive doubles (staged in LDS, 12 KB);
//   * products and the sum over s = 0..n_t-1 run in binary64 in the oracle's order (mul, add; no
//     contraction), rounded once to float -- bit-identical to oracle/ecc_oracle.c (eccor_ramp_filter);
//     MI355X runs vector fp64 at half the fp32 rate, so exactness is free at this size;
//   * in place: the row is in LDS before the first store and no other workgroup touches it.
#include <hip/hip_runtime.h>

#include "ecc_layout.h"

namespace {

constexpr int RAMP_THREADS = 256;

// OUT: outputs per thread per pass (t, t + 256, ...): 3 covers n_t <= 768 in one pass without idle slots (the default
// 768 bins), 4 is the general form.  h2 is staged in LDS next to the row: read from global memory, the four 512-byte
// loads per wave and step made the kernel L1-bound at 3.5x its float64 issue time (126 us per 768^2 dtr; now LDS-bound
// at 1.7x).
// H_LDS = false (more than 3276 distance bins: table + row above 64 KB): the table is read from global memory.
template <int OUT, bool H_LDS>
__global__ __launch_bounds__(RAMP_THREADS) void ramp_kernel(float* __restrict__ slabs, int64_t slab_stride,
                                                             int n_alpha, int n_t, int pitch,
                                                             const double* __restrict__ h2)
{
    extern __shared__ double lds_ramp[];  // [2 n_t doubles (h2),] then n_t floats (the row)
    const double* hs = H_LDS ? lds_ramp : h2;
    float* xs = reinterpret_cast<float*>(lds_ramp + (H_LDS ? 2 * n_t : 0));
    float* row = slabs + (int64_t)blockIdx.y * slab_stride + (size_t)(blockIdx.x + 1) * pitch + 1;
    for (int s = threadIdx.x; s < n_t; s += RAMP_THREADS) xs[s] = row[s];
    if (H_LDS)
        for (int s = threadIdx.x; s < 2 * n_t; s += RAMP_THREADS) lds_ramp[s] = h2[s];
    __syncthreads();
    for (int t0 = threadIdx.x; t0 < n_t; t0 += RAMP_THREADS * OUT) {
        double acc[OUT];
        const double* hp[OUT];
#pragma unroll
        for (int j = 0; j < OUT; ++j) {
            acc[j] = 0.0;
            // outputs past the end of the row read a valid (clamped) part of h2 and are not stored
            const int t = min(t0 + j * RAMP_THREADS, n_t - 1);
            hp[j] = hs + t + n_t;
        }
        for (int s = 0; s < n_t; ++s) {
            const double x = (double)xs[s];
#pragma unroll
            for (int j = 0; j < OUT; ++j) acc[j] += x * hp[j][-s];
        }
#pragma unroll
        for (int j = 0; j < OUT; ++j) {
            const int t = t0 + j * RAMP_THREADS;
            if (t < n_t) row[t] = (float)acc[j];
        }
    }
}

}  // namespace

// slabs: n_img dtrs in the private layout holding plain line integrals; h2_d: 2*n_t doubles.
// The caller re-runs the border replication afterwards (ecc_launch_dtr_border).
extern "C" hipError_t ecc_launch_ramp(float* slabs, int64_t slab_stride, int n_img, int n_alpha, int n_t, int pitch,
                                      const double* h2_d, hipStream_t stream)
{
    dim3 grid(n_alpha, n_img), block(RAMP_THREADS);
    const size_t lds_all = sizeof(double) * 2 * (size_t)n_t + sizeof(float) * (size_t)n_t;  // 15 KB at 768 bins
    if (n_t <= 3 * RAMP_THREADS)
        hipLaunchKernelGGL((ramp_kernel<3, true>), grid, block, lds_all, stream, slabs, slab_stride, n_alpha, n_t, pitch, h2_d);
    else if (lds_all <= 64 * 1024)
        hipLaunchKernelGGL((ramp_kernel<4, true>), grid, block, lds_all, stream, slabs, slab_stride, n_alpha, n_t, pitch, h2_d);
    else
        hipLaunchKernelGGL((ramp_kernel<4, false>), grid, block, sizeof(float) * (size_t)n_t, stream, slabs, slab_stride,
                           n_alpha, n_t, pitch, h2_d);
    return hipGetLastError();
}
