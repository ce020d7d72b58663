// Micro-benchmark: issue cost of VALU instruction classes on gfx950 (throughput with 8 waves/SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_t __attribute__((ext_vector_type(2)));
#define N_IT 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a0, float b0)
{
    float x0 = threadIdx.x * 1e-3f + a0, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    float x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    float2_t p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    const float2_t bb = {b0, b0 * 1.0001f};
    for (int i = 0; i < N_IT; ++i) {
        if (MODE == 0) {  // 8 scalar fma
            x0 = fmaf(x0, b0, 1.f); x1 = fmaf(x1, b0, 1.f); x2 = fmaf(x2, b0, 1.f); x3 = fmaf(x3, b0, 1.f);
            x4 = fmaf(x4, b0, 1.f); x5 = fmaf(x5, b0, 1.f); x6 = fmaf(x6, b0, 1.f); x7 = fmaf(x7, b0, 1.f);
        } else if (MODE == 1) {  // 4 packed fma (same flops)
            p0 = __builtin_elementwise_fma(p0, bb, bb); p1 = __builtin_elementwise_fma(p1, bb, bb);
            p2 = __builtin_elementwise_fma(p2, bb, bb); p3 = __builtin_elementwise_fma(p3, bb, bb);
        } else if (MODE == 2) {  // 8 rcp
            x0 = __builtin_amdgcn_rcpf(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_amdgcn_rcpf(x2); x3 = __builtin_amdgcn_rcpf(x3);
            x4 = __builtin_amdgcn_rcpf(x4); x5 = __builtin_amdgcn_rcpf(x5); x6 = __builtin_amdgcn_rcpf(x6); x7 = __builtin_amdgcn_rcpf(x7);
        } else if (MODE == 3) {  // 8 integer mul_lo
            int y0 = __float_as_int(x0) * 3201, y1 = __float_as_int(x1) * 3203, y2 = __float_as_int(x2) * 3205, y3 = __float_as_int(x3) * 3207;
            int y4 = __float_as_int(x4) * 3209, y5 = __float_as_int(x5) * 3211, y6 = __float_as_int(x6) * 3213, y7 = __float_as_int(x7) * 3215;
            x0 = __int_as_float(y0); x1 = __int_as_float(y1); x2 = __int_as_float(y2); x3 = __int_as_float(y3);
            x4 = __int_as_float(y4); x5 = __int_as_float(y5); x6 = __int_as_float(y6); x7 = __int_as_float(y7);
        } else if (MODE == 4) {  // 8 floor
            x0 = floorf(x0) + 0.5f; x1 = floorf(x1) + .5f; x2 = floorf(x2) + .5f; x3 = floorf(x3) + .5f;
            x4 = floorf(x4) + .5f; x5 = floorf(x5) + .5f; x6 = floorf(x6) + .5f; x7 = floorf(x7) + .5f;
        } else if (MODE == 5) {  // 8 cndmask + cmp
            x0 = x0 > b0 ? x1 : x2; x1 = x1 > b0 ? x2 : x3; x2 = x2 > b0 ? x3 : x4; x3 = x3 > b0 ? x4 : x5;
            x4 = x4 > b0 ? x5 : x6; x5 = x5 > b0 ? x6 : x7; x6 = x6 > b0 ? x7 : x0; x7 = x7 > b0 ? x0 : x1;
        } else if (MODE == 7) {  // 8 fract (v_fract_f32) + add
#define F(x) x = __builtin_amdgcn_fractf(x) + 1.5f
            F(x0); F(x1); F(x2); F(x3); F(x4); F(x5); F(x6); F(x7);
#undef F
        } else if (MODE == 8) {  // 8 rndne + add
#define F(x) x = __builtin_rintf(x) + .25f
            F(x0); F(x1); F(x2); F(x3); F(x4); F(x5); F(x6); F(x7);
#undef F
        } else if (MODE == 9) {  // 8 trunc + add
#define F(x) x = __builtin_truncf(x) + .25f
            F(x0); F(x1); F(x2); F(x3); F(x4); F(x5); F(x6); F(x7);
#undef F
        } else if (MODE == 10) {  // 8 cvt_i32_f32 + cvt_f32_i32
#define F(x) x = (float)((int)x + 1)
            F(x0); F(x1); F(x2); F(x3); F(x4); F(x5); F(x6); F(x7);
#undef F
        } else if (MODE == 11) {  // 16 add (the reference for the +add of the modes above)
#define F(x) x = (x + .25f) + b0
            F(x0); F(x1); F(x2); F(x3); F(x4); F(x5); F(x6); F(x7);
#undef F
        } else if (MODE == 6) {  // 8 f64 add
            double d0 = x0, d1 = x1; d0 += d1; d1 += d0; d0 += d1; d1 += d0; d0 += d1; d1 += d0; d0 += d1; d1 += d0;
            x0 = (float)d0; x1 = (float)d1;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE> void run(const char* name, float* out)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 0.999f);
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 0.999f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    // wave-iterations per SIMD = blocks*4 waves / 1024 SIMDs * N_IT ; cycles per wave-iteration on the SIMD pipe
    double cyc = ms * 1e-3 * 2.4e9 / ((double)blocks * 4 / 1024 * N_IT);
    printf("%-28s %8.3f ms  -> %.2f cycles per wave-iteration per SIMD (at 2.4 GHz)\n", name, ms, cyc);
}
int main()
{
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("8 x v_fma_f32", out);
    run<1>("4 x v_pk_fma_f32", out);
    run<2>("8 x v_rcp_f32", out);
    run<3>("8 x v_mul_lo_u32", out);
    run<4>("8 x (v_floor + v_add)", out);
    run<5>("8 x (v_cmp + v_cndmask)", out);
    run<6>("8 x v_add_f64 (+cvt)", out);
    run<7>("8 x (v_fract + v_add)", out);
    run<8>("8 x (v_rndne + v_add)", out);
    run<9>("8 x (v_trunc + v_add)", out);
    run<10>("8 x (cvt_i32 + add + cvt_f32)", out);
    run<11>("16 x v_add_f32", out);
    return 0;
}
