// Micro-benchmark: host-visible latency of ONE kernel launch whose first workgroup stores a flag into pinned host memory
// at once, by grid size, kernel-argument size, dynamic LDS and a per-workgroup read of pinned host memory.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_latency scripts/micro/launch_latency.hip && /tmp/launch_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>

struct Big { unsigned long long* flag; unsigned* ticket; const int* host_list; int mode; int pad; float blob[320]; };
struct Small { unsigned long long* flag; unsigned* ticket; const int* host_list; int mode; int pad; };

template <class A>
__global__ __launch_bounds__(1024) void k(A a, unsigned long long token)
{
    extern __shared__ float lds[];
    // mode 0: block 0 stores the flag at once.  mode 1: last-arriver (ticket) stores it.  mode 2: every block first reads 16
    // bytes of pinned host memory (like the index list), then the ticket.
    __shared__ unsigned t;
    int v = 0;
    if (a.mode == 2 && threadIdx.x < 4) v = a.host_list[4 * blockIdx.x + threadIdx.x];
    if (threadIdx.x == 0) lds[0] = (float)v;
    __syncthreads();
    if (a.mode == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0)
            __hip_atomic_store(a.flag, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if (threadIdx.x == 0) t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (t == gridDim.x - 1 && threadIdx.x == 0) {
        __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.flag, token + (lds[0] > 1e30f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class A>
double run(int grid, size_t lds, int mode, unsigned long long* flag_h, unsigned long long* flag_d, unsigned* ticket, const int* list_d,
           hipStream_t s, int idle_us, int threads = 256)
{
    A a;
    std::memset(&a, 0, sizeof(a));
    a.flag = flag_d; a.ticket = ticket; a.host_list = list_d; a.mode = mode;
    double tot = 0;
    const int reps = 200;
    for (int r = 0; r < reps + 20; ++r) {
        const unsigned long long token = 1000 + r;
        const double w = now();
        while (now() - w < idle_us * 1e-6) {}  // the host's share between two evaluations: the GPU idles
        const double t0 = now();
        hipLaunchKernelGGL((k<A>), dim3(grid), dim3(threads), lds, s, a, token);
        while (*(volatile unsigned long long*)flag_h != token) {}
        if (r >= 20) tot += now() - t0;
    }
    return 1e6 * tot / reps;
}

int main()
{
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned long long *flag_h, *flag_d; hipHostMalloc((void**)&flag_h, 64, hipHostMallocMapped); hipHostGetDevicePointer((void**)&flag_d, flag_h, 0);
    int *list_h, *list_d; hipHostMalloc((void**)&list_h, 16 * 4096, hipHostMallocMapped); hipHostGetDevicePointer((void**)&list_d, list_h, 0);
    std::memset(list_h, 0, 16 * 4096);
    unsigned* ticket; hipMalloc((void**)&ticket, 4); hipMemset(ticket, 0, 4);
    *flag_h = 0;
    for (int idle : {0, 10, 30}) {
        printf("host idles %d us between launches\n", idle);
        for (int grid : {1, 64, 399, 512, 1008, 4096}) {
            printf("  grid %4d: small args, first block stores %6.1f us | ticket %6.1f | + 6 KB LDS %6.1f | 1.4 KB args %6.1f | + host list read %6.1f\n", grid,
                   run<Small>(grid, 0, 0, flag_h, flag_d, ticket, list_d, s, idle), run<Small>(grid, 0, 1, flag_h, flag_d, ticket, list_d, s, idle),
                   run<Small>(grid, 6144, 1, flag_h, flag_d, ticket, list_d, s, idle), run<Big>(grid, 6144, 1, flag_h, flag_d, ticket, list_d, s, idle),
                   run<Big>(grid, 6144, 2, flag_h, flag_d, ticket, list_d, s, idle));
        }
    }
    printf("workgroup size at a fixed number of waves (ticket form, small arguments, no LDS):\n");
    for (int waves : {1024, 4096, 8192, 16384})
        printf("  %5d waves: 64 threads per workgroup %6.1f us | 256 %6.1f | 512 %6.1f | 1024 %6.1f\n", waves,
               run<Small>(waves, 0, 1, flag_h, flag_d, ticket, list_d, s, 0, 64), run<Small>(waves / 4, 0, 1, flag_h, flag_d, ticket, list_d, s, 0, 256),
               run<Small>(waves / 8, 0, 1, flag_h, flag_d, ticket, list_d, s, 0, 512), run<Small>(waves / 16, 0, 1, flag_h, flag_d, ticket, list_d, s, 0, 1024));
    return 0;
}
