// Micro-benchmark: cost of a wave-level gather instruction out of the L1 (TCP) on gfx950, by width and lane pattern.
// Every wave issues 4 independent gathers per trip (like the pair kernel) into a 20 KB working set.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_rate scripts/micro/gather_rate.hip && /tmp/gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>

struct __attribute__((packed, aligned(4))) F4 { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) F2 { float x, y; };
struct F1 { float x; };
#define N_IT 2048
__device__ __forceinline__ float total(const F4& v) { return (v.x + v.y) + (v.z + v.w); }
__device__ __forceinline__ float total(const F2& v) { return v.x + v.y; }
__device__ __forceinline__ float total(const F1& v) { return v.x; }

template <int PATTERN>
__device__ __forceinline__ unsigned lane_offset(int lane)
{
    switch (PATTERN) {
    case 0: return lane * 16;                                  // contiguous, 16-byte aligned
    case 1: return lane * 16 + 8;                              // contiguous, 8-byte aligned
    case 2: return lane * 8;                                   // footprints overlap by half
    case 3: return (lane * 3 / 8) * 8;                         // 0.375 bins per lane (the pair kernel's typical step)
    case 4: return (lane * 3 / 8) * 8 + (lane / 22) * 640;     // ... crossing three rows
    case 5: return 0;                                          // broadcast
    case 6: return ((lane * 2654435761u) >> 24) * 16;          // random 16-byte slots in 4 KB
    default: return ((lane * 2654435761u) >> 24) * 16 + 8;     // random, 8-byte aligned
    }
}

template <typename T, int PATTERN>
__global__ __launch_bounds__(256) void k(const char* __restrict__ buf, float* out)
{
    const int lane = threadIdx.x & 63;
    unsigned off = lane_offset<PATTERN>(lane);
    float acc = 0.f;
    for (int i = 0; i < N_IT; ++i) {
        asm volatile("" : "+v"(off));  // opaque: the loads cannot be hoisted or merged across trips
        const unsigned o = off + ((i & 3) << 8);  // four streams of ~5 KB each: the working set stays in the 32 KB L1
        const T a = *reinterpret_cast<const T*>(buf + o);
        const T b = *reinterpret_cast<const T*>(buf + o + 5120);
        const T c = *reinterpret_cast<const T*>(buf + o + 10240);
        const T d = *reinterpret_cast<const T*>(buf + o + 15360);
        acc += (total(a) + total(b)) + (total(c) + total(d));
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename T, int PATTERN> void run(const char* name, const char* buf, float* out)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL((k<T, PATTERN>), dim3(blocks), dim3(256), 0, 0, buf, out);
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<T, PATTERN>), dim3(blocks), dim3(256), 0, 0, buf, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    const double per_cu = (double)blocks * 4 * N_IT * 4 / 256;  // gather instructions per CU
    printf("%-44s %8.3f ms  -> %6.2f cycles per gather per CU (at 2.1 GHz), %5.1f B/clk/CU useful\n", name, ms,
           ms * 1e-3 * 2.1e9 / per_cu, 64.0 * sizeof(T) / (ms * 1e-3 * 2.1e9 / per_cu));
}

int main()
{
    char* buf; hipMalloc(&buf, 1 << 17); hipMemset(buf, 0, 1 << 17);
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<F4, 0>("dwordx4 contiguous aligned", buf, out);
    run<F4, 1>("dwordx4 contiguous 8-byte aligned", buf, out);
    run<F4, 2>("dwordx4 overlapping (8 B per lane)", buf, out);
    run<F4, 3>("dwordx4 0.375 bins per lane", buf, out);
    run<F4, 4>("dwordx4 0.375 bins per lane, 3 rows", buf, out);
    run<F4, 5>("dwordx4 broadcast", buf, out);
    run<F4, 6>("dwordx4 random aligned", buf, out);
    run<F4, 7>("dwordx4 random 8-byte aligned", buf, out);
    run<F2, 0>("dwordx2 contiguous (16 B stride)", buf, out);
    run<F2, 3>("dwordx2 0.375 bins per lane", buf, out);
    run<F2, 6>("dwordx2 random", buf, out);
    run<F1, 0>("dword (16 B stride)", buf, out);
    run<F1, 3>("dword 0.375 bins per lane", buf, out);
    run<F1, 6>("dword random", buf, out);
    return 0;
}
