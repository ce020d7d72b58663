// Micro-benchmark: does the LDS serve a 4-byte-aligned (odd word) ds_read_b64 correctly, and at which rate
// compared with ds_read2_b32 (the Radon kernel's bilinear row fetch) and an 8-byte-aligned ds_read_b64?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds64 scripts/micro/lds_b64_unaligned.hip && /tmp/lds64
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 4096
// MODE 0: ds_read2_b32 (words w, w+1)   1: ds_read_b64 at an even word   2: ds_read_b64 at an odd word
template <int MODE, int STRIDE>
__global__ __launch_bounds__(256) void k(float* out, int* bad)
{
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned word = (lane * STRIDE) & 4095;
    if (MODE == 1) word &= ~1u;
    if (MODE == 2) word |= 1u;
    unsigned addr = word * 4;
    float acc = 0.f;
    int wrong = 0;
    for (int i = 0; i < N_IT; ++i) {
        asm volatile("" : "+v"(addr));
        double d0, d1, d2, d3;  // four independent reads in flight, 8 KB apart (same banks)
        if (MODE == 0)
            asm volatile("ds_read2_b32 %0, %4 offset1:1\n ds_read2_b32 %1, %4 offset0:2 offset1:3\n"
                         "ds_read2_b32 %2, %4 offset0:4 offset1:5\n ds_read2_b32 %3, %4 offset0:6 offset1:7\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(addr));
        else
            asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %4 offset:16\n"
                         "ds_read_b64 %3, %4 offset:24\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(addr));
        const float a = __int_as_float((int)(__double_as_longlong(d0) & 0xffffffff));
        const float b = __int_as_float((int)(__double_as_longlong(d0) >> 32));
        const float c = __int_as_float((int)(__double_as_longlong(d3) >> 32));
        if (i == 0 && (a != (float)word || b != (float)(word + 1) || c != (float)(word + 7))) wrong++;
        acc += a + b + c + (float)d1 + (float)d2;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (wrong) atomicAdd(bad, 1);
}
template <int MODE, int STRIDE> void run(const char* name, float* out, int* bad)
{
    hipMemset(bad, 0, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 4;
    hipLaunchKernelGGL((k<MODE, STRIDE>), dim3(blocks), dim3(256), 0, 0, out, bad);
    hipEventRecord(a);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<MODE, STRIDE>), dim3(blocks), dim3(256), 0, 0, out, bad);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
    int nbad; hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost);
    const double per_cu = (double)blocks * 4 * N_IT * 4 / 256;
    printf("%-40s stride %2d words: %7.3f ms -> %5.2f cycles per wave read per CU (2.4 GHz)  %s\n", name, STRIDE, ms,
           ms * 1e-3 * 2.4e9 / per_cu, nbad ? "WRONG DATA" : "ok");
}
int main()
{
    float* out; hipMalloc(&out, 256 * 4 * 256 * 4);
    int* bad; hipMalloc(&bad, 4);
    run<0, 2>("ds_read2_b32 (w, w+1)", out, bad);
    run<1, 2>("ds_read_b64 8-byte aligned", out, bad);
    run<2, 2>("ds_read_b64 4-byte aligned (odd word)", out, bad);
    run<0, 3>("ds_read2_b32 (w, w+1)", out, bad);
    run<2, 3>("ds_read_b64 4-byte aligned (odd word)", out, bad);
    run<0, 97>("ds_read2_b32 (w, w+1)", out, bad);
    run<1, 97>("ds_read_b64 8-byte aligned", out, bad);
    run<2, 97>("ds_read_b64 4-byte aligned (odd word)", out, bad);
    return 0;
}
