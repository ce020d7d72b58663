// geometry_kernel.hip -- per-view pre-compute (E1) on the device.
//
// MetricRadonIntermediate::setProjectionMatrices (ref: LibEpipolarConsistency/
// EpipolarConsistencyRadonIntermediate.cpp:134-163) runs n small Householder QRs under OpenMP on
// the host and uploads 16 floats per view; inside an optimiser loop that is ~170 us of serial host
// work per evaluation for n = 400, a quarter of the pair kernel's time.  Here the caller's n x 12
// float64 matrices are uploaded (38 KB) and one thread per view performs exactly the same binary64
// arithmetic (ecc_host_geometry.h is compiled for both sides; add/mul/div/sqrt are correctly
// rounded and contraction is off), so PinvTs/Cs are bit-identical to the host result.
#include <hip/hip_runtime.h>

#include "ecc_host_geometry.h"

namespace {
// Two threads per view, one role per wave (no divergence): wave 0 of a workgroup computes (P^+)^T of 64 views, wave 1
// their source positions -- the two float64 Householder chains are independent and the kernel is pure latency (400
// views = a handful of waves), so splitting them shortens it (8 -> 6 us).
__global__ __launch_bounds__(128) void e1_kernel(const double* __restrict__ Ps, int n, float* __restrict__ PinvTs,
                                                 float* __restrict__ Cs)
{
    const int role = threadIdx.x >> 6;
    const int v = blockIdx.x * 64 + (threadIdx.x & 63);
    if (v >= n) return;
    double P[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) P[k] = Ps[12 * (size_t)v + k];
    if (role == 0) {
        float pinvT[12];
        ecc_host::pinv_transpose(P, pinvT);
#pragma unroll
        for (int k = 0; k < 12; ++k) PinvTs[12 * (size_t)v + k] = pinvT[k];
    } else {
        float C[4];
        ecc_host::source_position(P, C);
#pragma unroll
        for (int k = 0; k < 4; ++k) Cs[4 * (size_t)v + k] = C[k];
    }
}
}  // namespace

extern "C" hipError_t ecc_launch_e1(const double* Ps_d, int n, float* PinvTs_d, float* Cs_d, hipStream_t stream)
{
    hipLaunchKernelGGL(e1_kernel, dim3((n + 63) / 64), dim3(128), 0, stream, Ps_d, n, PinvTs_d, Cs_d);
    return hipGetLastError();
}
