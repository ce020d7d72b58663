// Micro-benchmark: one "evaluation" = four dependent kernels (7, 23, 50 or 330, 4 us) whose last one stores a flag into
// pinned host memory that the host polls -- issued as four stream launches or as one instantiated hipGraph.
// Prints us per evaluation for both forms (shard-sized and full-sized third kernel).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long cycles, volatile unsigned long long* flag, unsigned long long value)
{
    const long long t0 = wall_clock64();  // 100 MHz
    while (wall_clock64() - t0 < cycles) {}
    if (flag && threadIdx.x == 0 && blockIdx.x == 0) {
        __hip_atomic_store(const_cast<unsigned long long*>(flag), value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
int main()
{
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned long long* flag; hipHostMalloc((void**)&flag, 64, hipHostMallocMapped);
    volatile unsigned long long* vf = flag;
    const int pair_us[2] = {50, 330};
    for (int which = 0; which < 2; ++which) {
        const long long c[4] = {700, 2300, 100LL * pair_us[which], 400};
        const int reps = 300;
        // stream form
        double best[2] = {1e30, 1e30};
        for (int trial = 0; trial < 3; ++trial) {
            *vf = 0;
            double t0 = now_us();
            for (unsigned long long r = 1; r <= reps; ++r) {
                for (int k = 0; k < 4; ++k)
                    hipLaunchKernelGGL(spin, dim3(k == 2 ? 1024 : 64), dim3(256), 0, s, c[k], k == 3 ? flag : nullptr, r);
                while (*vf != r) {}
            }
            double t = (now_us() - t0) / reps;
            if (t < best[0]) best[0] = t;
        }
        // graph form (flag value fixed per graph: alternate two graphs storing 1 / 2)
        hipGraphExec_t ge[2];
        for (int g = 0; g < 2; ++g) {
            hipGraph_t graph;
            hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
            for (int k = 0; k < 4; ++k)
                hipLaunchKernelGGL(spin, dim3(k == 2 ? 1024 : 64), dim3(256), 0, s, c[k], k == 3 ? flag : nullptr, (unsigned long long)(g + 1));
            hipStreamEndCapture(s, &graph);
            hipGraphInstantiate(&ge[g], graph, nullptr, nullptr, 0);
            hipGraphDestroy(graph);
        }
        for (int trial = 0; trial < 3; ++trial) {
            hipStreamSynchronize(s);
            *vf = 0;
            double t0 = now_us();
            for (int r = 0; r < reps; ++r) {
                hipGraphLaunch(ge[r & 1], s);
                while (*vf != (unsigned long long)((r & 1) + 1)) {}
                *vf = 0;
            }
            double t = (now_us() - t0) / reps;
            if (t < best[1]) best[1] = t;
        }
        const double kernels = (c[0] + c[1] + c[2] + c[3]) / 100.0;
        printf("third kernel %3d us: kernels %.0f us; four stream launches %.1f us per evaluation, one graph launch %.1f us\n",
               pair_us[which], kernels, best[0], best[1]);
    }
    return 0;
}
