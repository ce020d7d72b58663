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
//     and every thread produces OUT consecutive outputs of that row;
//   * h comes from the host as a doubled table h2[m] = h[m mod n_t], m < 2 n_t, in binary64, so
//     h2[t - s + n_t] needs no modulo (staged in LDS, 12 KB);
//   * products and the sum over s = 0..n_t-1 run in binary64 in the oracle's order (mul, add; no
//     contraction), rounded once to float -- bit-identical to oracle/ecc_oracle.c (eccor_ramp_filter);
//     MI355X runs vector fp64 at half the fp32 rate, so exactness is free at this size;
//   * in place: the row is in LDS before the first store and no other workgroup touches it.
#include <hip/hip_runtime.h>

#include "ecc_layout.h"

namespace {

constexpr int RAMP_THREADS = 256;

// OUT: CONSECUTIVE outputs per thread (t, t + 1, ..., t + OUT - 1): 3 covers n_t <= 768 with one pass of 256 threads (the
// default 768 bins), 4 is the general form.  The kernel value an output needs moves by one table entry per step --
// h2[t + j - s + n_t] at step s is what output j - 1 used at step s - 1 -- so a thread keeps a window of OUT table values in
// registers and reads ONE new value per step (the window is a circular buffer whose rotation is unrolled away); with the
// outputs t, t + 256, t + 512 of rounds 1-4 every output read its own value every step and the kernel was bound by those
// LDS reads at 1.7x its float64 issue time (57 us per 768^2 dtr; round 2 with the table in global memory: 126 us).  The sum
// of every output still runs over s = 0 .. n_t - 1 in order, product then sum in binary64: the same bits.
// Lanes read table entries OUT doubles apart: conflict-free for OUT = 3 (3 is coprime to the 32 slots of ds_read_b64).
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
    const int last = 2 * n_t - 1;
    for (int base = 0; base < n_t; base += RAMP_THREADS * OUT) {
        const int t_own = base + (int)threadIdx.x * OUT;
        // a thread at the end of the row takes the last OUT outputs (some of them a second time: the same values); threads
        // past the end compute them too and store nothing
        const int t = max(0, min(t_own, n_t - OUT));
        double acc[OUT], w[OUT];
#pragma unroll
        for (int j = 0; j < OUT; ++j) {
            acc[j] = 0.0;
            w[j] = hs[min(t + j + n_t, last)];  // the window at s = 0
        }
        const double* next = hs + t + n_t - 1;  // what output 0 needs at step s + 1: next[-s]
        int s = 0;
        for (; s + OUT <= n_t; s += OUT) {
#pragma unroll
            for (int u = 0; u < OUT; ++u) {  // step s + u: output j uses slot (j - u) mod OUT
                const double x = (double)xs[s + u];
#pragma unroll
                for (int j = 0; j < OUT; ++j) acc[j] += x * w[(j - u + OUT) % OUT];
                w[(OUT - 1 - u + OUT) % OUT] = next[-(s + u)];  // the slot output OUT - 1 used becomes output 0's of the next step
            }
        }
        for (; s < n_t; ++s) {  // n_t not a multiple of OUT: the last steps with an explicit shift (slot j = output j again)
            const double x = (double)xs[s];
#pragma unroll
            for (int j = 0; j < OUT; ++j) acc[j] += x * w[j];
#pragma unroll
            for (int j = OUT - 1; j > 0; --j) w[j] = w[j - 1];
            w[0] = next[-s];
        }
        if (t_own < n_t) {
#pragma unroll
            for (int j = 0; j < OUT; ++j)
                if (t + j < n_t) row[t + j] = (float)acc[j];
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
