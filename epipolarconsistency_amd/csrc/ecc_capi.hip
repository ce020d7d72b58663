// ecc_capi.hip -- the C ABI declared in include/ecc_hip.h, host code: error state, the helpers every part shares (waits, the
// result slot, the fit's tables) and contexts.  The rest: ecc_radon_api.hip (Radon intermediates), ecc_metric_api.hip and
// ecc_evaluate.hip (MetricRadonIntermediate), ecc_preprocess_api.hip, ecc_direct_api.hip (MetricDirect), ecc_group.cpp.
//
// Host-side flow of the reference that this replaces:
//   RadonIntermediate ctor/compute   ref: LibEpipolarConsistency/RadonIntermediate.cpp:17-31,198-211
//   MetricRadonIntermediate::*        ref: LibEpipolarConsistency/EpipolarConsistencyRadonIntermediate.cpp
// Differences by design: one stream per context and no device-wide syncs between launches; K01 is
// fused into the pair kernel; the mean is reduced on the device in float64 and 8 bytes come back.
// There is NO CPU fallback: without a HIP device every compute entry point fails with
// ECC_ERR_NO_DEVICE / ECC_ERR_HIP.
#include "ecc_capi_internal.h"

#include <condition_variable>
#include <mutex>
#include <thread>

#define ECC_EXPORT extern "C" __attribute__((visibility("default")))

namespace {
thread_local std::string g_last_error;
}  // namespace

// (also for the other translation units of the library: ecc_capi_internal.h's fail(), ecc_group.cpp)
extern "C" int ecc_set_error(int code, const char* msg)
{
    g_last_error = msg ? msg : "";
    return code;
}

namespace ecc_internal {

// Wait for the stream with a query spin: the evaluate calls sit on an optimiser's critical path and the result is
// 8 bytes; hipStreamSynchronize's blocking wait costs several microseconds more per call than polling.  The spin is
// bounded: after ECC_SPIN_SECONDS of polling (a hung kernel, a faulted queue) it falls back to the blocking wait, which
// sleeps instead of burning a core and returns the queue's error when the driver gives up on it.
constexpr double ECC_SPIN_SECONDS = 2.0;

double now_seconds()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

hipError_t wait_stream_spin(hipStream_t stream)
{
    double t0 = 0.0;
    for (unsigned spins = 0;; ++spins) {
        const hipError_t e = hipStreamQuery(stream);
        if (e != hipErrorNotReady) return e;
        if ((spins & 0x3ff) == 0x3ff) {
            const double t = now_seconds();
            if (t0 == 0.0) t0 = t;
            else if (t - t0 > ECC_SPIN_SECONDS) return hipStreamSynchronize(stream);
        }
    }
}

// The result of an evaluation is one float64 that sum_pairs_kernel stores straight into pinned host memory.  The host
// arms the slot with a bit pattern no sum can have (a NaN with a private payload), launches, and polls the slot itself:
// the value is visible as soon as the store has crossed PCIe, a few microseconds before the runtime reports the stream
// idle (end-of-kernel cache write-back, completion signal).  Bounded like wait_stream_spin; the stream's own status is
// consulted when the value does not show up, so a failed launch surfaces as an error instead of a hang.
constexpr uint64_t ECC_RESULT_PENDING = 0x7ff8ecc0dead0001ull;

std::atomic<bool> g_result_polling{true};  // ecc_debug_set_result_polling(0): wait for the stream instead (A/B measurements)
bool result_polling_enabled() { return g_result_polling.load(std::memory_order_relaxed); }

void arm_result(ecc_metric* m)
{
    if (m->small_pending_count > 0) {
        // a synchronous call enqueued a one-launch evaluation and failed before it collected the result: that kernel still
        // stores its "done" word into this slot -- let it, before the slot is armed for the next evaluation
        (void)hipStreamSynchronize(m->ctx->stream);
        m->small_pending_count = 0;
    }
    reinterpret_cast<volatile uint64_t*>(m->sum_h)[0] = ECC_RESULT_PENDING;
    std::atomic_thread_fence(std::memory_order_seq_cst);
}

hipError_t wait_result(ecc_metric* m, hipStream_t stream, double* value)
{
    if (!result_polling_enabled()) {
        const hipError_t e = wait_stream_spin(stream);
        std::memcpy(value, m->sum_h, sizeof(double));
        return e;
    }
    const volatile uint64_t* slot = reinterpret_cast<const volatile uint64_t*>(m->sum_h);
    double t0 = 0.0;
    for (unsigned spins = 0;; ++spins) {
        const uint64_t bits = *slot;
        if (bits != ECC_RESULT_PENDING) {
            std::memcpy(value, &bits, sizeof(double));
            return hipSuccess;
        }
        if ((spins & 0xfff) == 0xfff) {
            const double t = now_seconds();
            if (t0 == 0.0) t0 = t;
            else if (t - t0 > ECC_SPIN_SECONDS) {
                const hipError_t e = hipStreamSynchronize(stream);
                const uint64_t b2 = *slot;
                std::memcpy(value, &b2, sizeof(double));
                if (e == hipSuccess && b2 == ECC_RESULT_PENDING) return hipErrorUnknown;  // the kernel ran and wrote nothing
                return e;
            }
        }
    }
}

int set_device(const ecc_ctx* ctx)
{
    HIP_TRY(hipSetDevice(ctx->device));
    return ECC_OK;
}

int ensure_poly_tables(ecc_ctx* ctx)
{
    if (ctx->poly_d) return ECC_OK;
    constexpr int N = ECC_POLY_DEG + 1;
    EccPolyTables t;
    for (int j = 0; j < N; ++j) t.nodes[j] = std::cos(3.14159265358979323846 * (j + 0.5) / N);
    const double checks[ECC_POLY_CHECKS] = {-1.0, 0.13, 1.0};  // both ends: largest interpolation error, and the fold state there
    for (int j = 0; j < ECC_POLY_CHECKS; ++j) t.checks[j] = checks[j];
    // inverse Vandermonde matrices in z = x^2 for the even part (nodes z_0..z_H, z_H = 0) and the odd part
    // (nodes z_0..z_{H-1}): Gauss-Jordan with partial pivoting in long double
    constexpr int H = ECC_POLY_DEG / 2;
    auto invert_vandermonde = [&](int n, double* out) {
        long double M[H + 1][2 * (H + 1)];
        for (int j = 0; j < n; ++j) {
            const long double z = (long double)t.nodes[j] * (long double)t.nodes[j];
            long double pw = 1;
            for (int k = 0; k < n; ++k) {
                M[j][k] = pw;
                pw *= z;
                M[j][n + k] = j == k ? 1 : 0;
            }
        }
        for (int col = 0; col < n; ++col) {
            int piv = col;
            for (int r = col + 1; r < n; ++r)
                if (fabsl(M[r][col]) > fabsl(M[piv][col])) piv = r;
            for (int k = 0; k < 2 * n; ++k) std::swap(M[col][k], M[piv][k]);
            const long double d = M[col][col];
            for (int k = 0; k < 2 * n; ++k) M[col][k] /= d;
            for (int r = 0; r < n; ++r) {
                if (r == col) continue;
                const long double f = M[r][col];
                for (int k = 0; k < 2 * n; ++k) M[r][k] -= f * M[col][k];
            }
        }
        for (int k = 0; k < n; ++k)
            for (int j = 0; j < n; ++j) out[k * n + j] = (double)M[k][n + j];
    };
    t.nodes[H] = 0.0;  // exactly
    invert_vandermonde(H + 1, t.Ae);
    invert_vandermonde(H, t.Ao);
    for (int k = 0; k <= ECC_TRIG_STEPS; ++k) {
        const double a = 3.14159265358979323846 * k / (2.0 * ECC_TRIG_STEPS);
        t.sc[k][0] = k == 0 ? 0.0 : k == ECC_TRIG_STEPS ? 1.0 : std::sin(a);
        t.sc[k][1] = k == 0 ? 1.0 : k == ECC_TRIG_STEPS ? 0.0 : std::cos(a);
    }
    HIP_TRY(hipMalloc((void**)&ctx->poly_d, sizeof(EccPolyTables)));
    HIP_TRY(hipMemcpyAsync(ctx->poly_d, &t, sizeof(t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // t goes out of scope
    return ECC_OK;
}

}  // namespace ecc_internal

using namespace ecc_internal;
// ---- misc ------------------------------------------------------------------------------------
ECC_EXPORT const char* ecc_last_error(void) { return g_last_error.c_str(); }

// for the other translation units of the library (ecc_exchange.hip)
ECC_EXPORT int ecc_version(void) { return ECC_HIP_VERSION; }
ECC_EXPORT int ecc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

ECC_EXPORT int ecc_host_line_to_sample_dtr(float* line, float range_t)
{
    // ref: EpipolarConsistencyCommon.hxx:152-171, the reference's float expressions; atan2 through binary64, rounded once
    const float Pi = 3.14159265359f;
    const float length = std::sqrt(line[0] * line[0] + line[1] * line[1]);
    line[0] = (float)std::atan2((double)line[1], (double)line[0]) / Pi;
    if (line[0] < 0) line[0] += 2;
    line[1] = -(line[2] / length) / range_t + 0.5f;
    if (line[0] > 1) {
        line[0] = line[0] - 1.f;
        line[1] = 1.f - line[1];
        return 1;
    }
    return 0;
}

ECC_EXPORT void ecc_get_ij(int64_t ij, int n, int* i, int* j)
{
    // pairs before row r: r*n - r(r+1)/2   (ref: EpipolarConsistencyCommon.hxx:52-79 enumerates the same order)
    double nn = (double)n - 0.5;
    int64_t r = (int64_t)std::floor(nn - std::sqrt(nn * nn - 2.0 * (double)ij));
    if (r < 0) r = 0;
    if (r > n - 2) r = n - 2;
    while (r > 0 && r * n - r * (r + 1) / 2 > ij) --r;
    while ((r + 1) * n - (r + 1) * (r + 2) / 2 <= ij) ++r;
    *i = (int)r;
    *j = (int)(ij - (r * n - r * (r + 1) / 2) + r + 1);
}
ECC_EXPORT void ecc_host_pinvT(const double* P, float* PinvT12) { ecc_host::pinv_transpose(P, PinvT12); }
ECC_EXPORT void ecc_host_source_position(const double* P, float* C4) { ecc_host::source_position(P, C4); }
ECC_EXPORT double ecc_host_object_radius(const double* P, int n_u, int n_v)
{
    return ecc_host::object_radius(P, n_u, n_v);
}

// ---- context -----------------------------------------------------------------------------------
ECC_EXPORT int ecc_ctx_create(int device, void* stream, ecc_ctx** out)
{
    if (!out) return fail(ECC_ERR_INVALID_ARGUMENT, "out is null");
    int n = ecc_device_count();
    if (n <= 0) return fail(ECC_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    if (device < 0 || device >= n) return fail(ECC_ERR_INVALID_ARGUMENT, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    ecc_ctx* c = new (std::nothrow) ecc_ctx();
    if (!c) return fail(ECC_ERR_OUT_OF_MEMORY, "host allocation failed");
    c->device = device;
    c->stream = (hipStream_t)stream;
    *out = c;
    return ECC_OK;
}

ECC_EXPORT int ecc_ctx_destroy(ecc_ctx* ctx)
{
    if (!ctx) return ECC_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->trig_d) (void)hipFree(ctx->trig_d);
    if (ctx->ramp_d) (void)hipFree(ctx->ramp_d);
    if (ctx->poly_d) (void)hipFree(ctx->poly_d);
    if (ctx->pre_tables_d) (void)hipFree(ctx->pre_tables_d);
    if (ctx->pre_tables_h) (void)hipHostFree(ctx->pre_tables_h);
    if (ctx->pre_ev) (void)hipEventDestroy(ctx->pre_ev);
    for (float* b : ctx->pre_scratch_d)
        if (b) (void)hipFree(b);
    if (ctx->radon_T_d) (void)hipFree(ctx->radon_T_d);
    if (ctx->linear_scratch_d) (void)hipFree(ctx->linear_scratch_d);
    for (auto& e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    delete ctx;
    return ECC_OK;
}

ECC_EXPORT int ecc_ctx_synchronize(ecc_ctx* ctx)
{
    if (!ctx) return fail(ECC_ERR_INVALID_ARGUMENT, "ctx is null");
    int rc = set_device(ctx);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ECC_OK;
}

ECC_EXPORT int ecc_debug_set_result_polling(int on)
{
    g_result_polling.store(on != 0, std::memory_order_relaxed);
    return ECC_OK;
}

ECC_EXPORT int ecc_ctx_set_quad_copies(ecc_ctx* ctx, int mode)
{
    if (!ctx) return fail(ECC_ERR_INVALID_ARGUMENT, "context is null");
    if (mode != ECC_QUAD_COPIES_AUTO && mode != ECC_QUAD_COPIES_OFF && mode != ECC_QUAD_COPIES_ON)
        return fail(ECC_ERR_INVALID_ARGUMENT, "quad-copy mode must be ECC_QUAD_COPIES_AUTO, _OFF or _ON");
    ctx->quad_copies = mode;
    return ECC_OK;
}

ECC_EXPORT int ecc_debug_set_quad_copies(ecc_ctx* ctx, int on) { return ecc_ctx_set_quad_copies(ctx, on ? ECC_QUAD_COPIES_ON : ECC_QUAD_COPIES_OFF); }

ECC_EXPORT int ecc_ctx_enable_timing(ecc_ctx* ctx, int enable)
{
    if (!ctx) return fail(ECC_ERR_INVALID_ARGUMENT, "ctx is null");
    int rc = set_device(ctx);
    if (rc) return rc;
    if (enable)
        for (auto& e : ctx->ev)
            if (!e) HIP_TRY(hipEventCreate(&e));
    ctx->timing = enable != 0;
    ctx->ev_valid[0] = ctx->ev_valid[1] = ctx->ev_valid[2] = false;
    return ECC_OK;
}

ECC_EXPORT int ecc_ctx_last_kernel_ms(ecc_ctx* ctx, int which, float* ms)
{
    if (!ctx || !ms || which < 0 || which > 2) return fail(ECC_ERR_INVALID_ARGUMENT, "bad argument");
    if (!ctx->timing || !ctx->ev_valid[which]) return fail(ECC_ERR_INVALID_ARGUMENT, "no timed launch recorded");
    int rc = set_device(ctx);
    if (rc) return rc;
    HIP_TRY(hipEventSynchronize(ctx->ev[2 * which + 1]));
    HIP_TRY(hipEventElapsedTime(ms, ctx->ev[2 * which], ctx->ev[2 * which + 1]));
    return ECC_OK;
}
