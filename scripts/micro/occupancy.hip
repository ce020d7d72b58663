#include <hip/hip_runtime.h>
#include <cstdio>
template <int KB> __global__ __launch_bounds__(256) void k(float* out)
{
    __shared__ float t[KB * 256];
    t[threadIdx.x] = threadIdx.x;
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = t[(threadIdx.x * 7) % (KB * 256)];
}
template <int KB> void q()
{
    int n = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<KB>, 256, 0);
    printf("static LDS %3d KB -> %d blocks of 256 threads per CU\n", KB, n);
}
int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    printf("%s: CUs %d, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, regsPerBlock %d, maxThreadsPerMP %d, clock %d kHz\n",
           p.name, p.multiProcessorCount, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock,
           p.maxThreadsPerMultiProcessor, p.clockRate);
    q<8>(); q<16>(); q<20>(); q<32>(); q<37>(); q<40>(); q<64>();
    return 0;
}
