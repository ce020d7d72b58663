// How many 256-thread workgroups (= waves per SIMD) does a CU of gfx950 hold as a function of the kernel's SGPR count?
// The pair kernel uses 106 scalar registers and runs with 6 waves per SIMD; the compiler's own estimate for it is 7.
// Each instantiation touches one high SGPR (so that the kernel's allocation reaches it), spins ~60 us so that every slot of
// the chip fills, and every wave records the wave-slot id of HW_REG_HW_ID: the largest id seen + 1 = resident waves per SIMD.
// hipcc --offload-arch=gfx950 -O2 scripts/micro/sgpr_occupancy.hip -o /tmp/sgpr_occ && /tmp/sgpr_occ
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define KERNEL(S, REG)                                                                                         \
    __global__ __launch_bounds__(256) void k##S(unsigned* out)                                                 \
    {                                                                                                          \
        asm volatile("s_mov_b32 " REG ", 0" ::: REG);                                                          \
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();                                        \
        while (__builtin_amdgcn_s_memrealtime() - t0 < 6000) __builtin_amdgcn_s_sleep(8);                      \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (15 << 11)); \
    }
KERNEL(48, "s47")
KERNEL(64, "s63")
KERNEL(72, "s71")
KERNEL(80, "s79")
KERNEL(88, "s87")
KERNEL(94, "s93")
KERNEL(96, "s95")
KERNEL(100, "s99")
KERNEL(102, "s101")

template <class K> void run(K kern, int S)
{
    const int blocks = 256 * 12;
    unsigned* d;
    (void)hipMalloc(&d, sizeof(unsigned) * blocks * 4);
    (void)hipMemset(d, 0xff, sizeof(unsigned) * blocks * 4);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(blocks * 4);
    (void)hipMemcpy(h.data(), d, sizeof(unsigned) * blocks * 4, hipMemcpyDeviceToHost);
    unsigned mx = 0;
    for (unsigned v : h) mx = (v & 0xf) > mx ? (v & 0xf) : mx;
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, 0);
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(kern));
    printf("touches s%-3d: runtime says %d workgroups per CU; largest wave slot seen %u -> %u waves per SIMD\n", S - 1, occ, mx, mx + 1);
    (void)hipFree(d);
}

int main()
{
    run(k48, 48); run(k64, 64); run(k72, 72); run(k80, 80); run(k88, 88); run(k94, 94); run(k96, 96); run(k100, 100); run(k102, 102);
    return 0;
}
