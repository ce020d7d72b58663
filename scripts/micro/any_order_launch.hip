// Micro-benchmark: does hipExtAnyOrderLaunch let a small kernel run BESIDE the previous kernel of the same stream on gfx950
// (AQL packet without the barrier bit), and does a normal launch behind both wait for both?  And what do the alternatives
// cost on the host: a launch on a second stream with fork / join events against one any-order launch on the same stream.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/any_order scripts/micro/any_order_launch.hip && /tmp/any_order
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

// every workgroup spins `ticks` of the 100 MHz wall clock; workgroup 0 records its start and end
__global__ __launch_bounds__(256) void spin(unsigned long long* out, long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while ((long long)(wall_clock64() - t0) < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = t0;
        out[1] = wall_clock64();
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    unsigned long long* d;
    CK(hipMalloc((void**)&d, 6 * sizeof(unsigned long long)));
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t fork_ev, join_ev;
    CK(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join_ev, hipEventDisableTiming));
    unsigned long long h[6];
    {   // what the individual runtime calls cost the host (stream idle; averages of 2000)
        const int N = 2000;
        double t = now();
        for (int i = 0; i < N; ++i) (void)hipStreamQuery(s);
        const double q = (now() - t) / N;
        t = now();
        for (int i = 0; i < N; ++i) CK(hipEventRecord(fork_ev, s));
        const double r = (now() - t) / N;
        CK(hipDeviceSynchronize());
        t = now();
        for (int i = 0; i < N; ++i) CK(hipStreamWaitEvent(s2, fork_ev, 0));
        const double w = (now() - t) / N;
        CK(hipDeviceSynchronize());
        t = now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 0ll);
            CK(hipStreamSynchronize(s));
        }
        const double l = (now() - t) / N;
        t = now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 0ll);
        const double l2 = (now() - t) / N;
        CK(hipDeviceSynchronize());
        t = now();
        for (int i = 0; i < N; ++i) (void)hipSetDevice(0);
        const double sd = (now() - t) / N;
        printf("host cost per call: hipStreamQuery (idle) %.2f us, hipEventRecord %.2f, hipStreamWaitEvent %.2f, launch + hipStreamSynchronize %.2f, "
               "launch alone (queue filling) %.2f, hipSetDevice %.2f\n", 1e6 * q, 1e6 * r, 1e6 * w, 1e6 * l, 1e6 * l2, 1e6 * sd);
    }
    for (int big = 0; big < 2; ++big) {
        const dim3 gridA(big ? 16384 : 8);  // 8 workgroups: the chip is nearly empty; 16384: every wave slot taken for a while
        const long long ticksA = big ? 1500 : 5000;  // 15 us per workgroup (several rounds) / 50 us
        for (int mode = 0; mode < 3; ++mode) {
            // mode 0: A, B, C all in order on one stream.  1: B with hipExtAnyOrderLaunch.  2: B on a second stream with fork / join events.
            double host_us = 0;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemset(d, 0, sizeof(h)));
                CK(hipDeviceSynchronize());
                const double t0 = now();
                hipLaunchKernelGGL(spin, gridA, dim3(256), 0, s, d, ticksA);
                const double t1 = now();
                if (mode == 0) hipLaunchKernelGGL(spin, dim3(100), dim3(256), 0, s, d + 2, 300ll);
                else if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(100), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d + 2, 300ll);
                else {
                    CK(hipEventRecord(fork_ev, s));  // (the library records it before A; here: what the calls cost)
                    CK(hipStreamWaitEvent(s2, fork_ev, 0));
                    hipLaunchKernelGGL(spin, dim3(100), dim3(256), 0, s2, d + 2, 300ll);
                    CK(hipEventRecord(join_ev, s2));
                    CK(hipStreamWaitEvent(s, join_ev, 0));
                }
                const double t2 = now();
                hipLaunchKernelGGL(spin, dim3(1), dim3(256), 0, s, d + 4, 100ll);
                CK(hipGetLastError());
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
                if (rep == 4)
                    printf("%s A, mode %d (%s): A %.1f..%.1f us, B %.1f..%.1f, C %.1f..%.1f   | host: launch A %.1f us, B path %.1f us\n",
                           big ? "chip-filling" : "small", mode, mode == 0 ? "in order" : mode == 1 ? "B any-order, same stream" : "B on a second stream, fork/join events",
                           0.0, (h[1] - h[0]) / 100.0, ((long long)h[2] - (long long)h[0]) / 100.0, ((long long)h[3] - (long long)h[0]) / 100.0,
                           ((long long)h[4] - (long long)h[0]) / 100.0, ((long long)h[5] - (long long)h[0]) / 100.0, 1e6 * (t1 - t0), 1e6 * (t2 - t1));
                host_us += 1e6 * (t2 - t1);
            }
        }
    }
    return 0;
}
