// ecc_capi.hip -- implementation of the C ABI declared in include/ecc_hip.h (host code only).
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

namespace {

int ensure_trig(ecc_ctx* ctx, int n_alpha)
{
    if (ctx->trig_d && ctx->trig_n_alpha == n_alpha) return ECC_OK;
    if (ctx->trig_d) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipFree(ctx->trig_d));
        ctx->trig_d = nullptr;
    }
    // alpha of angle bin ix, ref: RadonIntermediate.cu:46-50 (fp32, same expressions); the sine and
    // cosine are taken once per angle on the host instead of once per thread on the device, correctly
    // rounded (binary64 evaluation rounded once) so that the table does not depend on the libm.
    const float Pi = 3.14159265359f;
    std::vector<float> t(2 * (size_t)n_alpha);
    for (int ix = 0; ix < n_alpha; ++ix) {
        float x_rel = (ix / (float)n_alpha - 0.5f);
        float alpha = x_rel * Pi;
        t[2 * ix] = (float)std::sin((double)alpha);
        t[2 * ix + 1] = (float)std::cos((double)alpha);
    }
    HIP_TRY(hipMalloc((void**)&ctx->trig_d, t.size() * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(ctx->trig_d, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // t goes out of scope
    ctx->trig_n_alpha = n_alpha;
    return ECC_OK;
}

// Filter::Ramp as a circular convolution: h2[m] = h[m mod n_t], h[m] = sum_k w_k cos(2 pi k m / n_t),
// w_k = (float)min(k, n_t-k) * scale with the reference's float scale -0.5f/(n_t*n_theta)
// (ref: RadonIntermediate.cu:173-183,219); binary64, same expressions as oracle/ecc_oracle.c.
int ensure_ramp(ecc_ctx* ctx, int n_t)
{
    if (ctx->ramp_d && ctx->ramp_n_t == n_t) return ECC_OK;
    if (ctx->ramp_d) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipFree(ctx->ramp_d));
        ctx->ramp_d = nullptr;
    }
    const int n_theta = n_t / 2 + 1;
    const float scale = -0.5f / (n_t * n_theta);
    std::vector<double> c((size_t)n_t), h2(2 * (size_t)n_t);
    for (int r = 0; r < n_t; ++r) c[r] = std::cos(6.283185307179586476925286766559 * (double)r / (double)n_t);
    for (int m = 0; m < n_t; ++m) {
        double acc = 0.0;
        for (int k = 0; k < n_t; ++k) {
            const int kk = k <= n_t - k ? k : n_t - k;
            const float w = kk * scale;
            acc += (double)w * c[((long long)k * m) % n_t];
        }
        h2[m] = h2[(size_t)m + n_t] = acc;
    }
    HIP_TRY(hipMalloc((void**)&ctx->ramp_d, h2.size() * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(ctx->ramp_d, h2.data(), h2.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // h2 goes out of scope
    ctx->ramp_n_t = n_t;
    return ECC_OK;
}

// Chebyshev nodes, check abscissae and the inverse Vandermonde matrix of the per-pair polynomial fit
// (pairs_kernel.hip, fit_sample_polynomials); float64, built once per context.
int radon_launch(ecc_ctx* ctx, const float* images_d, int n, int n_u, int n_v, int n_alpha, int n_t, int filter,
                 int post, float* slabs, int64_t slab_stride)
{
    int rc = ensure_trig(ctx, n_alpha);
    if (rc) return rc;
    if (filter == ECC_FILTER_RAMP) {
        rc = ensure_ramp(ctx, n_t);
        if (rc) return rc;
    }
    // Workgroups whose lines run closer to x than to y (normal closer to y) stage their LDS tile transposed, from a
    // transposed copy of the images (radon_kernel.hip): one extra pass over the stack (8 bytes per pixel, ~2 us per
    // 1024^2 image against ~700 us of Radon kernel).  The copy is scratch in the context, at most RADON_SUB images
    // (256 MB at 1024^2) at a time; larger batches are launched in sub-batches on the same stream.
    constexpr int RADON_SUB = 64;
    const int64_t img_floats = (int64_t)n_u * n_v;
    const int sub = std::min(n, RADON_SUB);
    if (ctx->radon_T_cap < (size_t)img_floats * sub) {
        if (ctx->radon_T_d) {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            HIP_TRY(hipFree(ctx->radon_T_d));
            ctx->radon_T_d = nullptr;
            ctx->radon_T_cap = 0;
        }
        HIP_TRY(hipMalloc((void**)&ctx->radon_T_d, sizeof(float) * (size_t)img_floats * sub));
        ctx->radon_T_cap = (size_t)img_floats * sub;
    }
    if (ctx->timing) HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
    for (int first = 0; first < n; first += sub) {
        const int cnt = std::min(sub, n - first);
        EccRadonParams p;
        p.images = images_d + img_floats * first;
        p.imagesT = ctx->radon_T_d;
        p.out = slabs + slab_stride * first;
        p.trig = ctx->trig_d;
        p.image_stride = img_floats;
        p.out_stride = slab_stride;
        p.n_img = cnt;
        p.n_u = n_u;
        p.n_v = n_v;
        p.n_alpha = n_alpha;
        p.n_t = n_t;
        p.pitch = ecc_layout_pitch(n_t);
        p.post_process = post;
        p.arithmetic = ctx->radon_arithmetic;
        HIP_TRY(ecc_launch_direct_transpose(p.images, ctx->radon_T_d, cnt, n_u, n_v, ctx->stream));
        HIP_TRY(ecc_launch_radon(&p, filter == ECC_FILTER_DERIVATIVE ? 1 : 0, ctx->stream));
    }
    const int pitch = ecc_layout_pitch(n_t);
    if (filter == ECC_FILTER_RAMP) {
        // ref: RadonIntermediate.cu:166-167 (apply1DRampFilter after the plain line integrals)
        HIP_TRY(ecc_launch_ramp(slabs, slab_stride, n, n_alpha, n_t, pitch, ctx->ramp_d, ctx->stream));
        HIP_TRY(ecc_launch_dtr_border(slabs, slab_stride, n, n_alpha, n_t, pitch, ctx->stream));
    }
    if (ctx->timing) {
        HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
        ctx->ev_valid[1] = true;
    }
    return ECC_OK;
}

}  // namespace

ECC_EXPORT int ecc_radon_set_arithmetic(ecc_ctx* ctx, int mode)
{
    if (!ctx) return fail(ECC_ERR_INVALID_ARGUMENT, "context is null");
    if (mode != ECC_RADON_EXACT && mode != ECC_RADON_FMA) return fail(ECC_ERR_INVALID_ARGUMENT, "unknown Radon arithmetic mode");
    ctx->radon_arithmetic = mode;
    return ECC_OK;
}

ECC_EXPORT int ecc_radon_get_arithmetic(const ecc_ctx* ctx, int* mode)
{
    if (!ctx || !mode) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    *mode = ctx->radon_arithmetic;
    return ECC_OK;
}

namespace {

int check_radon_args(ecc_ctx* ctx, const float* image, int n, int n_u, int n_v, int n_alpha, int n_t, int filter,
                     int post, ecc_dtr** out)
{
    if (!ctx || !image || !out) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (n <= 0 || n > 65535) return fail(ECC_ERR_INVALID_ARGUMENT, "batch size must be in [1, 65535]");
    if (n_u < 2 || n_v < 2 || n_u > 16384 || n_v > 16384)
        return fail(ECC_ERR_INVALID_ARGUMENT, "image size must be in [2, 16384]");
    if (n_alpha < 1 || n_t < 1 || n_alpha > 16384 || n_t > 16384)
        return fail(ECC_ERR_INVALID_ARGUMENT, "Radon bin counts must be in [1, 16384]");
    if (filter != ECC_FILTER_DERIVATIVE && filter != ECC_FILTER_RAMP && filter != ECC_FILTER_NONE)
        return fail(ECC_ERR_INVALID_ARGUMENT, "unknown filter");
    if (post < 0 || post > 2) return fail(ECC_ERR_INVALID_ARGUMENT, "unknown post-process");
    return ECC_OK;
}

}  // namespace

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

ECC_EXPORT int ecc_debug_set_quad_copies(ecc_ctx* ctx, int on)
{
    if (!ctx) return fail(ECC_ERR_INVALID_ARGUMENT, "context is null");
    ctx->quad_copies = on != 0;
    return ECC_OK;
}

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

// ---- Radon intermediate ------------------------------------------------------------------------
ECC_EXPORT int64_t ecc_dtr_slab_floats(int n_alpha, int n_t) { return ecc_layout_floats(n_alpha, n_t); }

ECC_EXPORT int ecc_radon_compute_batch(ecc_ctx* ctx, const float* images, int images_on_device, int n, int n_u,
                                       int n_v, int n_alpha, int n_t, int filter, int post_process, ecc_dtr** out)
{
    int rc = check_radon_args(ctx, images, n, n_u, n_v, n_alpha, n_t, filter, post_process, out);
    if (rc) return rc;
    rc = set_device(ctx);
    if (rc) return rc;
    const int64_t slab = ecc_layout_floats(n_alpha, n_t);
    auto owner = std::make_shared<Slab>();
    owner->device = ctx->device;
    HIP_TRY(hipMalloc((void**)&owner->ptr, (size_t)slab * n * sizeof(float)));
    HIP_TRY(hipMemsetAsync(owner->ptr, 0, (size_t)slab * n * sizeof(float), ctx->stream));
    const float* images_d = images;
    float* staging = nullptr;
    if (!images_on_device) {
        size_t bytes = (size_t)n * n_u * n_v * sizeof(float);
        HIP_TRY(hipMalloc((void**)&staging, bytes));
        hipError_t e = hipMemcpyAsync(staging, images, bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) {
            (void)hipFree(staging);
            HIP_TRY(e);
        }
        images_d = staging;
    }
    rc = radon_launch(ctx, images_d, n, n_u, n_v, n_alpha, n_t, filter, post_process, owner->ptr, slab);
    if (staging) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(staging);
    }
    if (rc) return rc;
    for (int k = 0; k < n; ++k) {
        ecc_dtr* d = new (std::nothrow) ecc_dtr();
        if (!d) {
            for (int q = 0; q < k; ++q) delete out[q];
            return fail(ECC_ERR_OUT_OF_MEMORY, "host allocation failed");
        }
        d->ctx = ctx;
        d->owner = owner;
        d->base = owner->ptr + slab * k;
        d->n_alpha = n_alpha;
        d->n_t = n_t;
        d->n_u = n_u;
        d->n_v = n_v;
        d->filter = filter;
        d->pitch = ecc_layout_pitch(n_t);
        out[k] = d;
    }
    return ECC_OK;
}

ECC_EXPORT int ecc_radon_compute_into(ecc_ctx* ctx, const float* images_d, int n, int n_u, int n_v, int n_alpha,
                                      int n_t, int filter, int post_process, float* slabs_d)
{
    ecc_dtr* dummy = nullptr;
    int rc = check_radon_args(ctx, images_d, n, n_u, n_v, n_alpha, n_t, filter, post_process, &dummy);
    if (rc) return rc;
    if (!slabs_d) return fail(ECC_ERR_INVALID_ARGUMENT, "slabs_d is null");
    rc = set_device(ctx);
    if (rc) return rc;
    const int64_t slab = ecc_layout_floats(n_alpha, n_t);
    HIP_TRY(hipMemsetAsync(slabs_d, 0, (size_t)slab * n * sizeof(float), ctx->stream));
    return radon_launch(ctx, images_d, n, n_u, n_v, n_alpha, n_t, filter, post_process, slabs_d, slab);
}

ECC_EXPORT int ecc_radon_compute(ecc_ctx* ctx, const float* image, int image_on_device, int n_u, int n_v,
                                 int n_alpha, int n_t, int filter, int post_process, ecc_dtr** out)
{
    return ecc_radon_compute_batch(ctx, image, image_on_device, 1, n_u, n_v, n_alpha, n_t, filter, post_process, out);
}

ECC_EXPORT int ecc_dtr_from_host(ecc_ctx* ctx, const float* data, int n_alpha, int n_t, int n_u, int n_v, int filter,
                                 ecc_dtr** out)
{
    if (!ctx || !data || !out) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (n_alpha < 1 || n_t < 1 || n_alpha > 16384 || n_t > 16384)
        return fail(ECC_ERR_INVALID_ARGUMENT, "Radon bin counts must be in [1, 16384]");
    int rc = set_device(ctx);
    if (rc) return rc;
    const int64_t slab = ecc_layout_floats(n_alpha, n_t);
    auto owner = std::make_shared<Slab>();
    owner->device = ctx->device;
    HIP_TRY(hipMalloc((void**)&owner->ptr, (size_t)slab * sizeof(float)));
    HIP_TRY(hipMemsetAsync(owner->ptr, 0, (size_t)slab * sizeof(float), ctx->stream));
    float* staging = nullptr;
    size_t bytes = (size_t)n_alpha * n_t * sizeof(float);
    HIP_TRY(hipMalloc((void**)&staging, bytes));
    hipError_t e = hipMemcpyAsync(staging, data, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = ecc_launch_dtr_import(staging, owner->ptr, n_alpha, n_t, ecc_layout_pitch(n_t), ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(staging);
    HIP_TRY(e);
    ecc_dtr* d = new (std::nothrow) ecc_dtr();
    if (!d) return fail(ECC_ERR_OUT_OF_MEMORY, "host allocation failed");
    d->ctx = ctx;
    d->owner = owner;
    d->base = owner->ptr;
    d->n_alpha = n_alpha;
    d->n_t = n_t;
    d->n_u = n_u;
    d->n_v = n_v;
    d->filter = filter;
    d->pitch = ecc_layout_pitch(n_t);
    *out = d;
    return ECC_OK;
}

// The reference's launcher seam (ref: RadonIntermediate.cpp:12, RadonIntermediate.cu:149-170): image and result in device
// memory owned by the caller, the result in the REFERENCE's layout -- n_t rows of n_alpha floats, angle fastest, exactly
// n_t * n_alpha floats (what RadonIntermediate::compute allocates, ref: RadonIntermediate.cpp:208, and readback copies
// verbatim, :148-163).  Computed in a scratch slab of the context and transposed out; stream-ordered.
ECC_EXPORT int ecc_radon_compute_linear(ecc_ctx* ctx, const float* image_d, int n_u, int n_v, int n_alpha, int n_t, int filter,
                                        int post_process, float* out_linear_d)
{
    ecc_dtr* dummy = nullptr;
    int rc = check_radon_args(ctx, image_d, 1, n_u, n_v, n_alpha, n_t, filter, post_process, &dummy);
    if (rc) return rc;
    if (!out_linear_d) return fail(ECC_ERR_INVALID_ARGUMENT, "out_linear_d is null");
    rc = set_device(ctx);
    if (rc) return rc;
    const int64_t slab = ecc_layout_floats(n_alpha, n_t);
    if (ctx->linear_scratch_cap < (size_t)slab) {
        if (ctx->linear_scratch_d) {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            HIP_TRY(hipFree(ctx->linear_scratch_d));
            ctx->linear_scratch_d = nullptr;
            ctx->linear_scratch_cap = 0;
        }
        HIP_TRY(hipMalloc((void**)&ctx->linear_scratch_d, sizeof(float) * (size_t)slab));
        ctx->linear_scratch_cap = (size_t)slab;
    }
    HIP_TRY(hipMemsetAsync(ctx->linear_scratch_d, 0, (size_t)slab * sizeof(float), ctx->stream));
    rc = radon_launch(ctx, image_d, 1, n_u, n_v, n_alpha, n_t, filter, post_process, ctx->linear_scratch_d, slab);
    if (rc) return rc;
    HIP_TRY(ecc_launch_dtr_export(ctx->linear_scratch_d, out_linear_d, n_alpha, n_t, ecc_layout_pitch(n_t), ctx->stream));
    return ECC_OK;
}

// A Radon intermediate from DEVICE memory in the reference's layout (n_t x n_alpha, angle fastest): what the reference
// turns into a texture (ref: RadonIntermediate.cpp:188-196 getTexture: a copy into a cudaArray -- a snapshot, like here).
ECC_EXPORT int ecc_dtr_from_device_linear(ecc_ctx* ctx, const float* data_d, int n_alpha, int n_t, int n_u, int n_v, int filter,
                                          ecc_dtr** out)
{
    if (!ctx || !data_d || !out) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (n_alpha < 1 || n_t < 1 || n_alpha > 16384 || n_t > 16384)
        return fail(ECC_ERR_INVALID_ARGUMENT, "Radon bin counts must be in [1, 16384]");
    int rc = set_device(ctx);
    if (rc) return rc;
    const int64_t slab = ecc_layout_floats(n_alpha, n_t);
    auto owner = std::make_shared<Slab>();
    owner->device = ctx->device;
    HIP_TRY(hipMalloc((void**)&owner->ptr, (size_t)slab * sizeof(float)));
    HIP_TRY(hipMemsetAsync(owner->ptr, 0, (size_t)slab * sizeof(float), ctx->stream));
    HIP_TRY(ecc_launch_dtr_import(data_d, owner->ptr, n_alpha, n_t, ecc_layout_pitch(n_t), ctx->stream));
    ecc_dtr* d = new (std::nothrow) ecc_dtr();
    if (!d) return fail(ECC_ERR_OUT_OF_MEMORY, "host allocation failed");
    d->ctx = ctx;
    d->owner = owner;
    d->base = owner->ptr;
    d->n_alpha = n_alpha;
    d->n_t = n_t;
    d->n_u = n_u;
    d->n_v = n_v;
    d->filter = filter;
    d->pitch = ecc_layout_pitch(n_t);
    *out = d;
    return ECC_OK;
}

ECC_EXPORT int ecc_dtr_wrap_device(ecc_ctx* ctx, float* base, int n_alpha, int n_t, int n_u, int n_v, int filter,
                                   ecc_dtr** out)
{
    if (!ctx || !base || !out) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (n_alpha < 1 || n_t < 1) return fail(ECC_ERR_INVALID_ARGUMENT, "bad Radon bin counts");
    ecc_dtr* d = new (std::nothrow) ecc_dtr();
    if (!d) return fail(ECC_ERR_OUT_OF_MEMORY, "host allocation failed");
    d->ctx = ctx;
    d->base = base;
    d->n_alpha = n_alpha;
    d->n_t = n_t;
    d->n_u = n_u;
    d->n_v = n_v;
    d->filter = filter;
    d->pitch = ecc_layout_pitch(n_t);
    *out = d;
    return ECC_OK;
}

ECC_EXPORT int ecc_dtr_readback(ecc_dtr* dtr, float* host_out)
{
    if (!dtr || !host_out) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    ecc_ctx* ctx = dtr->ctx;
    int rc = set_device(ctx);
    if (rc) return rc;
    float* staging = nullptr;
    size_t bytes = (size_t)dtr->n_alpha * dtr->n_t * sizeof(float);
    HIP_TRY(hipMalloc((void**)&staging, bytes));
    hipError_t e = ecc_launch_dtr_export(dtr->base, staging, dtr->n_alpha, dtr->n_t, dtr->pitch, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(host_out, staging, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(staging);
    HIP_TRY(e);
    return ECC_OK;
}

ECC_EXPORT int ecc_dtr_info(const ecc_dtr* dtr, int* n_alpha, int* n_t, int* n_u, int* n_v, int* filter,
                            double* bin_size_angle, double* bin_size_distance)
{
    if (!dtr) return fail(ECC_ERR_INVALID_ARGUMENT, "dtr is null");
    if (n_alpha) *n_alpha = dtr->n_alpha;
    if (n_t) *n_t = dtr->n_t;
    if (n_u) *n_u = dtr->n_u;
    if (n_v) *n_v = dtr->n_v;
    if (filter) *filter = dtr->filter;
    // ref: RadonIntermediate.cpp:204-206
    if (bin_size_angle) *bin_size_angle = 3.1415926535897931 / dtr->n_alpha;
    if (bin_size_distance)
        *bin_size_distance = std::sqrt((double)dtr->n_v * dtr->n_v + (double)dtr->n_u * dtr->n_u) / dtr->n_t;
    return ECC_OK;
}

ECC_EXPORT int ecc_dtr_device_view(const ecc_dtr* dtr, float** base, int* pitch, int* rows)
{
    if (!dtr) return fail(ECC_ERR_INVALID_ARGUMENT, "dtr is null");
    if (base) *base = dtr->base;
    if (pitch) *pitch = dtr->pitch;
    if (rows) *rows = ecc_layout_rows(dtr->n_alpha);
    return ECC_OK;
}

ECC_EXPORT int ecc_dtr_destroy(ecc_dtr* dtr)
{
    if (!dtr) return ECC_OK;
    if (dtr->owner && dtr->owner.use_count() == 1) {
        (void)hipSetDevice(dtr->ctx->device);
        (void)hipStreamSynchronize(dtr->ctx->stream);
    }
    delete dtr;
    return ECC_OK;
}

// ---- metric ------------------------------------------------------------------------------------
ECC_EXPORT int ecc_metric_create(ecc_ctx* ctx, int n_dtrs, ecc_dtr* const* dtrs, ecc_metric** out)
{
    if (!ctx || !dtrs || !out) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (n_dtrs < 1) return fail(ECC_ERR_INVALID_ARGUMENT, "need at least one Radon intermediate");
    for (int k = 0; k < n_dtrs; ++k)
        if (!dtrs[k]) return fail(ECC_ERR_INVALID_ARGUMENT, "null Radon intermediate in list");
    int rc = set_device(ctx);
    if (rc) return rc;
    ecc_metric* m = new (std::nothrow) ecc_metric();
    if (!m) return fail(ECC_ERR_OUT_OF_MEMORY, "host allocation failed");
    m->ctx = ctx;
    {
        const char* e = std::getenv("ECC_RECORD_REUSE");  // 0: off, 1: default, 2: for every size (include/ecc_hip.h, next to ecc_metric_set_record_reuse)
        if (e && e[0] >= '0' && e[0] <= '2') m->record_reuse = e[0] - '0';
    }
    m->dtrs.assign(dtrs, dtrs + n_dtrs);
    // sizes come from dtrs[0] only, ref: ...RadonIntermediate.cpp:92-98
    const ecc_dtr* d0 = dtrs[0];
    m->n_alpha = d0->n_alpha;
    m->n_t = d0->n_t;
    m->n_u = d0->n_u;
    m->n_v = d0->n_v;
    m->pitch = d0->pitch;
    m->is_derivative = d0->filter == ECC_FILTER_DERIVATIVE;
    m->step_alpha = (float)(3.1415926535897931 / d0->n_alpha);
    m->step_t = (float)(std::sqrt((double)d0->n_v * d0->n_v + (double)d0->n_u * d0->n_u) / d0->n_t);
    std::vector<const float*> table(n_dtrs);
    for (int k = 0; k < n_dtrs; ++k) {
        // unlike the reference (mixed sizes are "silently wrong") reject them
        if (dtrs[k]->n_alpha != m->n_alpha || dtrs[k]->n_t != m->n_t) {
            delete m;
            return fail(ECC_ERR_INVALID_ARGUMENT, "all Radon intermediates must have the same bin counts");
        }
        table[k] = dtrs[k]->base;
    }
    // (the pair kernel forms byte offsets inside a paired copy in fp32 while the copy stays below 2^24 bytes and in
    // integer arithmetic above -- fill_pair_params; offsets have to fit 32 bits: 16384 x 16384 bins is 2.1 GB)
    const int64_t paired_floats = (int64_t)(m->n_alpha + 1) * m->pitch * 2;
    if (paired_floats * 4 >= (int64_t)1 << 32) {
        delete m;
        return fail(ECC_ERR_UNSUPPORTED, "Radon intermediates above 4 GB per row-paired copy are not supported");
    }
    std::vector<const float*> ptable(n_dtrs);
    hipError_t e = hipMalloc((void**)&m->dtr_table_d, sizeof(float*) * n_dtrs);
    if (e == hipSuccess) e = hipMalloc((void**)&m->paired_table_d, sizeof(float*) * n_dtrs);
    if (e == hipSuccess) e = hipMalloc((void**)&m->paired_d, sizeof(float) * (size_t)paired_floats * n_dtrs);
    for (int k = 0; k < n_dtrs && e == hipSuccess; ++k) ptable[k] = m->paired_d + (size_t)paired_floats * k;
    if (e == hipSuccess)
        e = hipMemcpyAsync(m->paired_table_d, ptable.data(), sizeof(float*) * n_dtrs, hipMemcpyHostToDevice, ctx->stream);
    // row-quad copies: opt-in (ecc_debug_set_quad_copies on the context; 4x the slab memory, see pairs_kernel.hip); offsets
    // must fit 32 bits
    std::vector<const float*> qtable(n_dtrs);
    m->quad_floats = (int64_t)((m->n_alpha + 1 + 3) / 4) * m->pitch * 16;
    const bool want_quads = m->quad_floats * 4 < ((int64_t)1 << 32) && ctx->quad_copies;
    if (e == hipSuccess && want_quads) {
        e = hipMalloc((void**)&m->quads_table_d, sizeof(float*) * n_dtrs);
        if (e == hipSuccess) e = hipMalloc((void**)&m->quads_d, sizeof(float) * (size_t)m->quad_floats * n_dtrs);
        for (int k = 0; k < n_dtrs && e == hipSuccess; ++k) qtable[k] = m->quads_d + (size_t)m->quad_floats * k;
        if (e == hipSuccess)
            e = hipMemcpyAsync(m->quads_table_d, qtable.data(), sizeof(float*) * n_dtrs, hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) e = hipMalloc((void**)&m->sum_d, sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&m->sum_scratch_d, ecc_sum_scratch_bytes());
    if (e == hipSuccess) e = hipMemsetAsync(m->sum_scratch_d, 0, ecc_sum_scratch_bytes(), ctx->stream);

    if (e == hipSuccess) e = hipHostMalloc((void**)&m->sum_h, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&m->sum_h_dev, m->sum_h, 0);
    if (e == hipSuccess)
        e = hipMemcpyAsync(m->dtr_table_d, table.data(), sizeof(float*) * n_dtrs, hipMemcpyHostToDevice, ctx->stream);
    // the metric borrows the dtrs and they must not change during its lifetime (ref: ...RadonIntermediate.h:45), so
    // the paired copies are built once, here
    if (e == hipSuccess)
        e = ecc_launch_build_paired(m->dtr_table_d, m->paired_d, paired_floats, n_dtrs, m->n_alpha + 1, m->pitch, ctx->stream);
    if (e == hipSuccess && m->quads_d)
        e = ecc_launch_build_quad(m->dtr_table_d, m->quads_d, m->quad_floats, n_dtrs, m->n_alpha + 1, m->pitch, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        ecc_metric_destroy(m);
        HIP_TRY(e);
    }
    *out = m;
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_refresh_dtrs(ecc_metric* m, int first, int count)
{
    if (!m) return fail(ECC_ERR_INVALID_ARGUMENT, "metric is null");
    const int n = (int)m->dtrs.size();
    if (first < 0 || count < 0 || first > n || count > n - first) return fail(ECC_ERR_INVALID_ARGUMENT, "dtr range outside the metric's list");
    if (count == 0) return ECC_OK;
    m->cache_valid = false;
    int rc = set_device(m->ctx);
    if (rc) return rc;
    const int64_t paired_floats = (int64_t)(m->n_alpha + 1) * m->pitch * 2;
    // stream-ordered behind whatever produced the new slab contents on this stream, in front of the next evaluation
    HIP_TRY(ecc_launch_build_paired(m->dtr_table_d + first, m->paired_d + (size_t)paired_floats * first, paired_floats, count,
                                    m->n_alpha + 1, m->pitch, m->ctx->stream));
    if (m->quads_d)
        HIP_TRY(ecc_launch_build_quad(m->dtr_table_d + first, m->quads_d + (size_t)m->quad_floats * first, m->quad_floats, count,
                                      m->n_alpha + 1, m->pitch, m->ctx->stream));
    return ECC_OK;
}

namespace ecc_internal {
// E1 on the device for the matrices of the last ecc_metric_set_projections, if nobody has run it yet: one thread per
// view reads its 12 doubles straight from the pinned staging buffer and does the reference's binary64 Householder-QR
// arithmetic (geometry_kernel.hip); nothing else crosses PCIe.
int ensure_e1(ecc_metric* m)
{
    if (!m->e1_pending) return ECC_OK;
    const int slot = (int)(m->set_generation & 1);
    const size_t n12 = (size_t)12 * m->n_views;
    if (m->dev_valid && m->dev_Ps.size() == n12 && std::memcmp(m->dev_Ps.data(), m->Ps_h[slot], sizeof(double) * n12) == 0) {
        m->e1_pending = false;  // the device arrays already belong to these matrices (patched view by view, or set back)
        return ECC_OK;
    }
    HIP_TRY(ecc_launch_e1(m->Ps_h_dev[slot], m->n_views, m->PinvTs_d, m->Cs_d, m->ctx->stream));
    m->dev_Ps.assign(m->Ps_h[slot], m->Ps_h[slot] + n12);
    m->dev_valid = true;
    m->e1_pending = false;
    // The reuse path of launch_range assumes PinvTs / Cs on the device are E1(rec_Ps) for every view it finds unchanged.
    // This launch has just made them E1 of the CURRENT matrices for all views (an image-pair or debug call between two
    // evaluations gets here), so the kept records no longer describe the device geometry: the next evaluation refits
    // everything.  (launch_range's own full refit comes through here too and sets rec_valid again when it is done.)
    m->rec_valid = false;
    return ECC_OK;
}
}  // namespace ecc_internal

ECC_EXPORT int ecc_metric_set_record_reuse(ecc_metric* m, int on)
{
    if (!m) return fail(ECC_ERR_INVALID_ARGUMENT, "metric is null");
    m->record_reuse = on < 0 ? 0 : (on > 2 ? 2 : on);
    m->rec_valid = false;
    if (!m->record_reuse) m->eager_e1 = true;  // until an evaluation says otherwise
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_set_small_eval(ecc_metric* m, int on)
{
    if (!m) return fail(ECC_ERR_INVALID_ARGUMENT, "metric is null");
    m->small_eval = on ? 1 : 0;
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_destroy(ecc_metric* m)
{
    if (!m) return ECC_OK;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    if (m->dtr_table_d) (void)hipFree((void*)m->dtr_table_d);
    if (m->paired_table_d) (void)hipFree((void*)m->paired_table_d);
    if (m->paired_d) (void)hipFree(m->paired_d);
    if (m->quads_table_d) (void)hipFree((void*)m->quads_table_d);
    if (m->quads_d) (void)hipFree(m->quads_d);
    if (m->Cs_d) (void)hipFree(m->Cs_d);
    if (m->PinvTs_d) (void)hipFree(m->PinvTs_d);
    if (m->pair_values_d) (void)hipFree(m->pair_values_d);
    if (m->cost_d) (void)hipFree(m->cost_d);
    if (m->indices_d) (void)hipFree(m->indices_d);
    if (m->K01_d) (void)hipFree(m->K01_d);
    if (m->records_d) (void)hipFree(m->records_d);
    if (m->sum_d) (void)hipFree(m->sum_d);
    if (m->sum_scratch_d) (void)hipFree(m->sum_scratch_d);
    if (m->Ps_d) (void)hipFree(m->Ps_d);
    for (double* b : m->Ps_h)
        if (b) (void)hipHostFree(b);
    if (m->sum_h) (void)hipHostFree(m->sum_h);
    if (m->cache_values_d) (void)hipFree(m->cache_values_d);
    if (m->list_h) (void)hipHostFree(m->list_h);
    for (int b = 0; b < 2; ++b) {
        if (m->reuse_h[b]) (void)hipHostFree(m->reuse_h[b]);
        if (m->reuse_ev[b]) (void)hipEventDestroy(m->reuse_ev[b]);
    }
    if (m->side_stream) {
        (void)hipStreamSynchronize(m->side_stream);
        (void)hipStreamDestroy(m->side_stream);
    }
    if (m->fork_ev) (void)hipEventDestroy(m->fork_ev);
    if (m->join_ev) (void)hipEventDestroy(m->join_ev);
    if (m->sidx_h) (void)hipHostFree(m->sidx_h);
    if (m->svals_h) (void)hipHostFree(m->svals_h);
    if (m->small_ticket_d) (void)hipFree(m->small_ticket_d);
    delete m;
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_set_projections(ecc_metric* m, const double* Ps, int n_views)
{
    if (!m || !Ps) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (n_views < 1) return fail(ECC_ERR_INVALID_ARGUMENT, "need at least one projection matrix");
    ecc_stamp(m, 0);
    ecc_ctx* ctx = m->ctx;
    int rc = set_device(ctx);
    if (rc) return rc;
    if (n_views > m->geom_capacity) {
        HIP_TRY(wait_stream_spin(ctx->stream));
        m->done_generation = m->set_generation;
        if (m->Cs_d) HIP_TRY(hipFree(m->Cs_d));
        if (m->PinvTs_d) HIP_TRY(hipFree(m->PinvTs_d));
        if (m->Ps_d) HIP_TRY(hipFree(m->Ps_d));
        for (double*& b : m->Ps_h) {
            if (b) HIP_TRY(hipHostFree(b));
            b = nullptr;
        }
        m->Cs_d = m->PinvTs_d = nullptr;
        m->Ps_d = nullptr;
        m->geom_capacity = 0;
        m->dev_valid = false;
        HIP_TRY(hipMalloc((void**)&m->Cs_d, sizeof(float) * 4 * n_views));
        HIP_TRY(hipMalloc((void**)&m->PinvTs_d, sizeof(float) * 12 * n_views));
        HIP_TRY(hipMalloc((void**)&m->Ps_d, sizeof(double) * 12 * n_views));
        for (int b = 0; b < 2; ++b) {
            HIP_TRY(hipHostMalloc((void**)&m->Ps_h[b], sizeof(double) * 12 * n_views, hipHostMallocMapped));
            HIP_TRY(hipHostGetDevicePointer((void**)&m->Ps_h_dev[b], m->Ps_h[b], 0));
        }
        m->geom_capacity = n_views;
    }
    // The staging buffer of this call was last read by the e1 launch two calls ago.  In the optimiser pattern
    // (setProjectionMatrices, evaluate, setProjectionMatrices, ...) that launch is known to be complete and nothing is
    // waited for; only a caller that sets matrices repeatedly without a synchronous evaluate in between waits here.
    const uint64_t g = m->set_generation + 1;
    if (g > 2 && m->done_generation < g - 2) {
        HIP_TRY(wait_stream_spin(ctx->stream));
        m->done_generation = m->set_generation;
    }
    const int slot = (int)(g & 1);
    std::memcpy(m->Ps_h[slot], Ps, sizeof(double) * 12 * (size_t)n_views);
    m->set_generation = g;
    m->n_views = n_views;
    m->P_first.assign(Ps, Ps + 12);
    // E1 itself is launched by whoever needs PinvTs / Cs next (ensure_e1): an evaluation that finds most matrices
    // unchanged computes the few changed views on the host and never launches it.
    m->e1_pending = true;
    if (m->eager_e1) rc = ensure_e1(m);  // the last evaluation needed it on the device and skipped nothing: launch it now
    ecc_stamp(m, 1);
    return rc;
}

/* Debug: read back what E1 produced on the device (12 + 4 floats per view). */
ECC_EXPORT int ecc_metric_debug_geometry(ecc_metric* m, float* PinvTs, float* Cs)
{
    if (!m || !PinvTs || !Cs) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (m->n_views < 1) return fail(ECC_ERR_INVALID_ARGUMENT, "projection matrices have not been set");
    int rc = set_device(m->ctx);
    if (rc) return rc;
    {
        const int rc1 = ensure_e1(m);
        if (rc1) return rc1;
    }
    HIP_TRY(hipMemcpyAsync(PinvTs, m->PinvTs_d, sizeof(float) * 12 * m->n_views, hipMemcpyDeviceToHost, m->ctx->stream));
    HIP_TRY(hipMemcpyAsync(Cs, m->Cs_d, sizeof(float) * 4 * m->n_views, hipMemcpyDeviceToHost, m->ctx->stream));
    HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_set_params(ecc_metric* m, double object_radius_mm, double dkappa, int use_corr)
{
    if (!m) return fail(ECC_ERR_INVALID_ARGUMENT, "metric is null");
    m->object_radius_mm = object_radius_mm;
    m->dkappa = dkappa;
    m->use_corr = use_corr;
    m->cache_valid = false;
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_set_sampling(ecc_metric* m, int mode)
{
    if (!m) return fail(ECC_ERR_INVALID_ARGUMENT, "metric is null");
    if (mode < ECC_SAMPLING_AUTO || mode > ECC_SAMPLING_REFERENCE) return fail(ECC_ERR_INVALID_ARGUMENT, "unknown sampling mode");
    m->sampling = mode;
    m->cache_valid = false;
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_set_incremental(ecc_metric* m, int enable)
{
    if (!m) return fail(ECC_ERR_INVALID_ARGUMENT, "metric is null");
    m->incremental = enable ? 1 : 0;
    m->cache_valid = false;
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_last_evaluated_pairs(const ecc_metric* m, int64_t* pairs)
{
    if (!m || !pairs) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    *pairs = m->last_evaluated_pairs;
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_get_object_radius(const ecc_metric* m, double* radius_mm)
{
    if (!m || !radius_mm) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (m->object_radius_mm > 0) *radius_mm = m->object_radius_mm;
    else if (m->P_first.empty()) *radius_mm = 0;
    else {
        if (!m->radius_cache_valid || std::memcmp(m->radius_cache_P, m->P_first.data(), sizeof(double) * 12) != 0) {
            m->radius_cache = ecc_host::object_radius(m->P_first.data(), m->n_u, m->n_v);
            std::memcpy(m->radius_cache_P, m->P_first.data(), sizeof(double) * 12);
            m->radius_cache_valid = true;
        }
        *radius_mm = m->radius_cache;
    }
    return ECC_OK;
}

// ---- evaluateForImagePair (E7) ------------------------------------------------------------------
ECC_EXPORT int ecc_metric_pair_samples_bound(const ecc_metric* m, int* capacity)
{
    if (!m || !capacity) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    // kappa runs over (-kappa_max, kappa_max) in steps of dkappa: automatic dkappa = 2 kappa_max / num_samples
    // gives num_samples steps; a user dkappa gives at most Pi / dkappa (kappa_max <= Pi/2).
    const float num_samples = sqrtf((float)(m->n_u * m->n_u + m->n_v * m->n_v));
    const double n = m->dkappa > 0 ? 3.14159265358979323846 / (double)(float)m->dkappa : (double)num_samples;
    if (!(n < 65536.0)) return fail(ECC_ERR_INVALID_ARGUMENT, "more than 65536 kappa samples (visualisation path)");
    *capacity = (int)n + 16;
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_evaluate_for_image_pair(ecc_metric* m, int i, int j, int capacity, int* n_samples,
                                                  float* rs0, float* rs1, float* kappas, float* radon0, float* radon1,
                                                  float* K01, double* ecc)
{
    if (!m || !n_samples) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (m->n_views < 1) return fail(ECC_ERR_INVALID_ARGUMENT, "projection matrices have not been set");
    const int nD = (int)m->dtrs.size();
    if (i < 0 || j < 0 || i >= m->n_views || j >= m->n_views || i >= nD || j >= nD)
        return fail(ECC_ERR_INVALID_ARGUMENT, "view index out of range");
    int bound = 0;
    int rc = ecc_metric_pair_samples_bound(m, &bound);
    if (rc) return rc;
    ecc_ctx* ctx = m->ctx;
    rc = set_device(ctx);
    if (rc) return rc;
    double radius = 0;
    ecc_metric_get_object_radius(m, &radius);

    float* out_d = nullptr;
    HIP_TRY(hipMalloc((void**)&out_d, sizeof(float) * (7 * (size_t)bound + 16) + sizeof(int)));
    float* K01_d = out_d + 7 * (size_t)bound;
    int* n_d = reinterpret_cast<int*>(K01_d + 16);
    EccPairSamplesParams p;
    std::memset(&p, 0, sizeof(p));
    p.dtr0 = m->dtrs[i]->base;
    p.dtr1 = m->dtrs[j]->base;
    rc = ensure_e1(m);
    if (rc) {
        (void)hipFree(out_d);
        return rc;
    }
    p.Cs = m->Cs_d;
    p.PinvTs = m->PinvTs_d;
    p.out = out_d;
    p.K01_out = K01_d;
    p.n_out = n_d;
    p.iP0 = i;
    p.iP1 = j;
    p.capacity = bound;
    p.n_alpha = m->n_alpha;
    p.n_t = m->n_t;
    p.pitch = m->pitch;
    p.n_x2 = m->n_u * 0.5f;
    p.n_y2 = m->n_v * 0.5f;
    p.object_radius_mm = (float)radius;
    p.num_samples = sqrtf((float)(m->n_u * m->n_u + m->n_v * m->n_v));  // ref: ...RadonIntermediate.cpp:349
    p.range_t = m->step_t * m->n_t;                                      // ref: RadonIntermediate.h:90
    p.dkappa_user = (float)m->dkappa;
    p.derivative0 = m->dtrs[i]->filter == ECC_FILTER_DERIVATIVE;
    p.derivative1 = m->dtrs[j]->filter == ECC_FILTER_DERIVATIVE;
    std::vector<float> host(7 * (size_t)bound + 16 + 1);
    hipError_t e = hipMemsetAsync(out_d, 0, sizeof(float) * (7 * (size_t)bound + 16) + sizeof(int), ctx->stream);
    if (e == hipSuccess) e = ecc_launch_pair_samples(&p, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(host.data(), out_d, sizeof(float) * host.size(), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(out_d);
    HIP_TRY(e);
    int n = 0;
    std::memcpy(&n, &host[7 * (size_t)bound + 16], sizeof(int));
    *n_samples = n;
    if (n >= bound) return fail(ECC_ERR_HIP, "internal: kappa sample bound exceeded");
    if (n > capacity) return fail(ECC_ERR_INVALID_ARGUMENT, "capacity is smaller than the number of kappa samples");
    const float* v0 = host.data();
    const float* v1 = v0 + bound;
    const float* kp = v1 + bound;
    const float* a0 = kp + bound;
    const float* d0 = a0 + bound;
    const float* a1 = d0 + bound;
    const float* d1 = a1 + bound;
    const float* K = d1 + bound;
    const float dkappa = K[8 + 6];
    double acc = 0;
    for (int k = 0; k < n; ++k) {
        if (rs0) rs0[k] = v0[k];
        if (rs1) rs1[k] = v1[k];
        if (kappas) kappas[k] = kp[k];
        if (radon0) { radon0[2 * k] = a0[k]; radon0[2 * k + 1] = d0[k]; }
        if (radon1) { radon1[2 * k] = a1[k]; radon1[2 * k + 1] = d1[k]; }
        acc += (double)((v0[k] - v1[k]) * (v0[k] - v1[k]) * dkappa);  // ref: ...RadonIntermediate.cpp:389, accumulated
    }
    if (K01) std::memcpy(K01, K, sizeof(float) * 16);
    if (ecc) *ecc = acc;
    return ECC_OK;
}

// ---- projection pre-processing --------------------------------------------------------------------
ECC_EXPORT void ecc_host_intrinsics(const double* P, float* sdd_px, float* ppu, float* ppv)
{
    ecc_host::intrinsics(P, sdd_px, ppu, ppv);
}

ECC_EXPORT void ecc_preprocess_defaults(ecc_preprocess_config* cfg)
{
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    // ref: Gui/PreProccess.h:19-45
    cfg->process = 1;
    cfg->scale = 1.0;
    cfg->gaussian_sigma = 1.84;
    cfg->half_kernel_width = 5;
    for (int s = 0; s < 4; ++s) {
        cfg->zero[s] = 1;
        cfg->feather[s] = 16;
    }
}

ECC_EXPORT int ecc_preprocess(ecc_ctx* ctx, const float* images, int on_device, float* out, int n, int n_u, int n_v,
                              const ecc_preprocess_config* cfg, const double* Ps)
{
    if (!ctx || !images || !out || !cfg) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (n <= 0 || n > 65535) return fail(ECC_ERR_INVALID_ARGUMENT, "batch size must be in [1, 65535]");
    if (n_u < 1 || n_v < 1 || n_u > 16384 || n_v > 16384)
        return fail(ECC_ERR_INVALID_ARGUMENT, "image size must be in [1, 16384]");
    if (cfg->n_blanks < 0 || (cfg->n_blanks > 0 && !cfg->blanks))
        return fail(ECC_ERR_INVALID_ARGUMENT, "bad blanks");
    for (int s = 0; s < 4; ++s)
        if (cfg->zero[s] < 0 || cfg->feather[s] < 0) return fail(ECC_ERR_INVALID_ARGUMENT, "negative border width");
    // ref: Gui/PreProccess.cpp:142: low-pass only if sigma > 0 and half width > 1
    const bool lowpass = cfg->process && cfg->gaussian_sigma > 0 && cfg->half_kernel_width > 1;
    const int k = lowpass ? cfg->half_kernel_width : 0;
    if (k > 16) return fail(ECC_ERR_UNSUPPORTED, "half kernel width above 16 is not supported");
    int rc = set_device(ctx);
    if (rc) return rc;

    const size_t img_floats = (size_t)n_u * n_v;
    // device-side tables: kernel (2k+1 doubles), blanks, per-image cosine-weight intrinsics, per-image maxima
    std::vector<double> kernel(2 * (size_t)k + 1, 0.0);
    if (k > 0) {  // ref: HeaderOnly/NRRD/nrrd_lowpass.hxx:19-33 (gaussianKernel)
        double sum = 0;
        for (int x = -k; x <= k; ++x) {
            const double v = std::exp(-0.5 * std::pow(x / cfg->gaussian_sigma, 2));
            sum += v;
            kernel[x + k] = v;
        }
        for (double& v : kernel) v /= sum;
    }
    std::vector<float> cosw;
    std::vector<int> valid;
    if (Ps) {
        cosw.resize(3 * (size_t)n);
        valid.resize(n);
        for (int v = 0; v < n; ++v) {
            const double* P = Ps + 12 * (size_t)v;
            bool zero = true;
            for (int e = 0; e < 12; ++e) zero = zero && P[e] == 0;
            valid[v] = zero ? 0 : 1;
            if (zero) cosw[3 * v] = cosw[3 * v + 1] = cosw[3 * v + 2] = 0.f;
            else ecc_host::intrinsics(P, &cosw[3 * v], &cosw[3 * v + 1], &cosw[3 * v + 2]);
        }
    }
    // border factors as tables over the source column / row (the kernel multiplies instead of re-deriving them per
    // pixel): ref Gui/PreProccess.cpp:86-113, same float / double expressions as the oracle's
    std::vector<float> border_w(2 * (size_t)n_u + 2 * (size_t)n_v, 1.0f);
    if (cfg->process) {
        auto weighting = [](double x) {  // ref: Gui/PreProccess.cpp:8-13
            if (x < -1.0 || x > 1.0) return 0.0;
            const double xx = x * x;
            return 1.0 - 2 * xx + xx * xx;
        };
        float *wl = border_w.data(), *wr = wl + n_u, *wb = wr + n_u, *wt = wb + n_v;
        const int* z = cfg->zero;
        const int* f = cfg->feather;
        for (int sx = 0; sx < n_u; ++sx) {
            if (sx < z[0] + f[0]) wl[sx] = sx <= z[0] ? 0.f : (float)weighting(1 - (float)(sx - z[0]) / f[0]);
            const int b = n_u - sx;
            if (b <= z[1] + f[1]) wr[sx] = b <= z[1] ? 0.f : (float)weighting(1 - (float)(b - z[1]) / f[1]);
        }
        for (int sy = 0; sy < n_v; ++sy) {
            const int b = n_v - sy;
            if (b <= z[2] + f[2]) wb[sy] = b <= z[2] ? 0.f : (float)weighting(1 - (float)(b - z[2]) / f[2]);
            if (sy < z[3] + f[3]) wt[sy] = sy <= z[3] ? 0.f : (float)weighting(1 - (float)(sy - z[3]) / f[3]);
        }
    }
    const size_t border_b = sizeof(float) * border_w.size();
    const size_t kernel_b = sizeof(double) * kernel.size();
    const size_t blanks_b = sizeof(int32_t) * 4 * (size_t)cfg->n_blanks;
    const size_t cosw_b = sizeof(float) * cosw.size(), valid_b = sizeof(int) * valid.size();
    const size_t max_b = sizeof(float) * (size_t)n * ECC_PRE_MAX_CHUNKS;
    auto up8 = [](size_t b) { return (b + 7) / 8 * 8; };
    const size_t upload_b = up8(kernel_b) + up8(blanks_b) + up8(cosw_b) + up8(valid_b) + up8(border_b);
    const size_t table_b = upload_b + up8(max_b);
    const size_t stack_b = sizeof(float) * img_floats * n;
    const bool in_place = on_device && images == out;
    if (on_device && !in_place) {
        // tiles read halos of their neighbours: a partially overlapping output would race with those reads
        const char *a0 = reinterpret_cast<const char*>(images), *b0 = reinterpret_cast<const char*>(out);
        if (a0 < b0 + stack_b && b0 < a0 + stack_b)
            return fail(ECC_ERR_INVALID_ARGUMENT, "out overlaps images without being identical to it");
    }
    // arena (kept in the context): tables, their pinned host image, scratch stacks
    if (ctx->pre_tables_cap < table_b) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->pre_tables_d) HIP_TRY(hipFree(ctx->pre_tables_d));
        if (ctx->pre_tables_h) HIP_TRY(hipHostFree(ctx->pre_tables_h));
        ctx->pre_tables_d = ctx->pre_tables_h = nullptr;
        ctx->pre_tables_cap = 0;
        const size_t cap = std::max(table_b * 2, (size_t)4096);
        HIP_TRY(hipMalloc((void**)&ctx->pre_tables_d, cap));
        HIP_TRY(hipHostMalloc((void**)&ctx->pre_tables_h, cap, hipHostMallocDefault));
        ctx->pre_tables_cap = cap;
        ctx->pre_ev_recorded = false;
    }
    if (!ctx->pre_ev) HIP_TRY(hipEventCreateWithFlags(&ctx->pre_ev, hipEventDisableTiming));
    auto ensure_scratch = [&](int which) -> int {
        if (ctx->pre_scratch_cap[which] >= stack_b) return ECC_OK;
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->pre_scratch_d[which]) HIP_TRY(hipFree(ctx->pre_scratch_d[which]));
        ctx->pre_scratch_d[which] = nullptr;
        ctx->pre_scratch_cap[which] = 0;
        HIP_TRY(hipMalloc((void**)&ctx->pre_scratch_d[which], stack_b));
        ctx->pre_scratch_cap[which] = stack_b;
        return ECC_OK;
    };
    float* staging_in = nullptr;   // host input
    float* staging_out = nullptr;  // host output, or the in-place form's scratch
    if (!on_device || in_place) {
        rc = ensure_scratch(0);
        if (rc) return rc;
        staging_out = ctx->pre_scratch_d[0];
    }
    if (!on_device) {
        rc = ensure_scratch(1);
        if (rc) return rc;
        staging_in = ctx->pre_scratch_d[1];
    }
    // the previous call's table upload may still be reading the pinned image
    if (ctx->pre_ev_recorded) HIP_TRY(hipEventSynchronize(ctx->pre_ev));
    char* th = ctx->pre_tables_h;
    if (kernel_b) std::memcpy(th, kernel.data(), kernel_b);
    th += up8(kernel_b);
    if (blanks_b) std::memcpy(th, cfg->blanks, blanks_b);
    th += up8(blanks_b);
    if (cosw_b) std::memcpy(th, cosw.data(), cosw_b);
    th += up8(cosw_b);
    if (valid_b) std::memcpy(th, valid.data(), valid_b);
    th += up8(valid_b);
    std::memcpy(th, border_w.data(), border_b);
    char* t = ctx->pre_tables_d;
    double* kernel_d = reinterpret_cast<double*>(t); t += up8(kernel_b);
    int* blanks_d = reinterpret_cast<int*>(t); t += up8(blanks_b);
    float* cosw_d = reinterpret_cast<float*>(t); t += up8(cosw_b);
    int* valid_d = reinterpret_cast<int*>(t); t += up8(valid_b);
    float* border_d = reinterpret_cast<float*>(t); t += up8(border_b);
    float* max_d = reinterpret_cast<float*>(t);
    hipError_t e = hipSuccess;
    if (upload_b) e = hipMemcpyAsync(ctx->pre_tables_d, ctx->pre_tables_h, upload_b, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipEventRecord(ctx->pre_ev, ctx->stream);
    if (e == hipSuccess) ctx->pre_ev_recorded = true;
    if (e == hipSuccess && !on_device)
        e = hipMemcpyAsync(staging_in, images, stack_b, hipMemcpyHostToDevice, ctx->stream);

    EccPreprocessParams p;
    std::memset(&p, 0, sizeof(p));
    p.in = on_device ? images : staging_in;
    p.out = (on_device && !in_place) ? out : staging_out;
    p.stride = (int64_t)img_floats;
    p.n_img = n;
    p.n_u = n_u;
    p.n_v = n_v;
    p.process = cfg->process ? 1 : 0;
    p.normalize = cfg->normalize ? 1 : 0;
    p.scale = (float)cfg->scale;  // ref: Gui/PreProccess.cpp:63-64
    p.bias = (float)cfg->bias;
    p.max_d = max_d;
    p.apply_log = cfg->apply_log ? 1 : 0;
    // the flips belong to PreProccess::process (ref: Gui/PreProccess.cpp:123-136); the cosine weighting alone leaves the
    // image where it is
    p.flip_u = (cfg->process && cfg->flip_u) ? 1 : 0;
    p.flip_v = (cfg->process && cfg->flip_v) ? 1 : 0;
    for (int s = 0; s < 4; ++s) {
        p.zero[s] = cfg->zero[s];
        p.feather[s] = cfg->feather[s];
    }
    p.n_blanks = cfg->n_blanks;
    p.blanks = blanks_d;
    p.k = k;
    p.kernel = kernel_d;
    p.cosw = Ps ? cosw_d : nullptr;
    p.cosw_valid = Ps ? valid_d : nullptr;
    p.border_w = border_d;
    if (ctx->timing && e == hipSuccess) e = hipEventRecord(ctx->ev[4], ctx->stream);
    if (e == hipSuccess) e = ecc_launch_preprocess(&p, ctx->stream);
    if (ctx->timing && e == hipSuccess) {
        e = hipEventRecord(ctx->ev[5], ctx->stream);
        ctx->ev_valid[2] = true;
    }
    if (e == hipSuccess && in_place)
        e = hipMemcpyAsync(out, staging_out, stack_b, hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess && !on_device) {
        e = hipMemcpyAsync(out, staging_out, stack_b, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // host output: the caller reads it next
    }
    // device forms (in place or not) are asynchronous on the context's stream: tables and scratch live in the context
    HIP_TRY(e);
    return ECC_OK;
}

// ---- MetricDirect ---------------------------------------------------------------------------------
struct ecc_direct {
    ecc_ctx* ctx = nullptr;
    int n_images = 0, n_u = 0, n_v = 0, n_views = 0;
    const float* images_d = nullptr;
    float* owned_images = nullptr;
    float* imagesT_d = nullptr;  // transposed copies (see direct_lines_kernel); refreshed by ecc_direct_update_images
    double object_radius_mm = 0, dkappa = 0;
    int use_fbcc = 0;
    std::vector<double> P_first;
    double* Ps_d = nullptr;
    EccDirectView* views_d = nullptr;
    int view_capacity = 0;
    // scratch, grown on demand
    EccDirectPair* pairs_d = nullptr;
    float* samples_d = nullptr;
    double* pair_metric_d = nullptr;
    int64_t batch_capacity = 0;
    int n_max_capacity = 0;
    double* total_d = nullptr;
    float* cost_d = nullptr;
    int cost_capacity = 0;
};

namespace {

int direct_n_max(const ecc_direct* d, int* n_max)
{
    // automatic dkappa = 0.5 * range / diagonal  ->  n_lines = (int)(range / dkappa) ~ 2 * diagonal;
    // a user dkappa gives at most Pi / dkappa lines (range <= Pi)            (ref: ...Direct.cpp:98-110)
    const double diag = std::sqrt((double)(d->n_u * d->n_u + d->n_v * d->n_v));
    const double n = d->dkappa > 0 ? 3.14159265358979323846 / d->dkappa : 2.0 * diag;
    if (!(n < 1048576.0)) return fail(ECC_ERR_INVALID_ARGUMENT, "more than 2^20 epipolar lines per pair");
    *n_max = (int)n + 4;
    return ECC_OK;
}

double direct_radius(const ecc_direct* d)
{
    // ref: Metric::getObjectRadius (EpipolarConsistency.cpp:76-84): user value or the FIRST view's estimate
    if (d->object_radius_mm > 0) return d->object_radius_mm;
    if (d->P_first.empty()) return 0;
    return ecc_host::object_radius(d->P_first.data(), d->n_u, d->n_v);
}

int direct_scratch(ecc_direct* d, int64_t batch, int n_max)
{
    if (d->batch_capacity >= batch && d->n_max_capacity >= n_max) return ECC_OK;
    HIP_TRY(hipStreamSynchronize(d->ctx->stream));
    if (d->pairs_d) (void)hipFree(d->pairs_d);
    if (d->samples_d) (void)hipFree(d->samples_d);
    if (d->pair_metric_d) (void)hipFree(d->pair_metric_d);
    d->pairs_d = nullptr; d->samples_d = nullptr; d->pair_metric_d = nullptr;
    d->batch_capacity = 0;
    HIP_TRY(hipMalloc((void**)&d->pairs_d, sizeof(EccDirectPair) * (size_t)batch));
    HIP_TRY(hipMalloc((void**)&d->samples_d, sizeof(float) * 2 * (size_t)batch * n_max));
    HIP_TRY(hipMalloc((void**)&d->pair_metric_d, sizeof(double) * (size_t)batch));
    d->batch_capacity = batch;
    d->n_max_capacity = n_max;
    return ECC_OK;
}

}  // namespace

ECC_EXPORT int ecc_direct_create(ecc_ctx* ctx, int n_images, const float* images, int on_device, int n_u, int n_v,
                                 ecc_direct** out)
{
    if (!ctx || !images || !out) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (n_images < 1) return fail(ECC_ERR_INVALID_ARGUMENT, "need at least one image");
    if (n_u < 2 || n_v < 2 || n_u > 16384 || n_v > 16384) return fail(ECC_ERR_INVALID_ARGUMENT, "image size must be in [2, 16384]");
    int rc = set_device(ctx);
    if (rc) return rc;
    ecc_direct* d = new (std::nothrow) ecc_direct();
    if (!d) return fail(ECC_ERR_OUT_OF_MEMORY, "host allocation failed");
    d->ctx = ctx;
    d->n_images = n_images;
    d->n_u = n_u;
    d->n_v = n_v;
    hipError_t e = hipMalloc((void**)&d->total_d, sizeof(double));
    if (e == hipSuccess && !on_device) {
        const size_t bytes = sizeof(float) * (size_t)n_images * n_u * n_v;
        e = hipMalloc((void**)&d->owned_images, bytes);
        if (e == hipSuccess) e = hipMemcpyAsync(d->owned_images, images, bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        d->images_d = d->owned_images;
    } else {
        d->images_d = images;
    }
    if (e == hipSuccess) e = hipMalloc((void**)&d->imagesT_d, sizeof(float) * (size_t)n_images * n_u * n_v);
    if (e == hipSuccess) e = ecc_launch_direct_transpose(d->images_d, d->imagesT_d, n_images, n_u, n_v, ctx->stream);
    if (e != hipSuccess) {
        ecc_direct_destroy(d);
        HIP_TRY(e);
    }
    *out = d;
    return ECC_OK;
}

ECC_EXPORT int ecc_direct_update_images(ecc_direct* d)
{
    if (!d) return fail(ECC_ERR_INVALID_ARGUMENT, "metric is null");
    int rc = set_device(d->ctx);
    if (rc) return rc;
    HIP_TRY(ecc_launch_direct_transpose(d->images_d, d->imagesT_d, d->n_images, d->n_u, d->n_v, d->ctx->stream));
    return ECC_OK;
}

ECC_EXPORT int ecc_direct_destroy(ecc_direct* d)
{
    if (!d) return ECC_OK;
    (void)hipSetDevice(d->ctx->device);
    (void)hipStreamSynchronize(d->ctx->stream);
    if (d->owned_images) (void)hipFree(d->owned_images);
    if (d->imagesT_d) (void)hipFree(d->imagesT_d);
    if (d->Ps_d) (void)hipFree(d->Ps_d);
    if (d->views_d) (void)hipFree(d->views_d);
    if (d->pairs_d) (void)hipFree(d->pairs_d);
    if (d->samples_d) (void)hipFree(d->samples_d);
    if (d->pair_metric_d) (void)hipFree(d->pair_metric_d);
    if (d->total_d) (void)hipFree(d->total_d);
    if (d->cost_d) (void)hipFree(d->cost_d);
    delete d;
    return ECC_OK;
}

ECC_EXPORT int ecc_direct_set_projections(ecc_direct* d, const double* Ps, int n_views)
{
    if (!d || !Ps) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (n_views < 1) return fail(ECC_ERR_INVALID_ARGUMENT, "need at least one projection matrix");
    ecc_ctx* ctx = d->ctx;
    int rc = set_device(ctx);
    if (rc) return rc;
    if (n_views > d->view_capacity) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (d->Ps_d) HIP_TRY(hipFree(d->Ps_d));
        if (d->views_d) HIP_TRY(hipFree(d->views_d));
        d->Ps_d = nullptr; d->views_d = nullptr; d->view_capacity = 0;
        HIP_TRY(hipMalloc((void**)&d->Ps_d, sizeof(double) * 12 * n_views));
        HIP_TRY(hipMalloc((void**)&d->views_d, sizeof(EccDirectView) * n_views));
        d->view_capacity = n_views;
    }
    HIP_TRY(hipMemcpyAsync(d->Ps_d, Ps, sizeof(double) * 12 * n_views, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ecc_launch_direct_views(d->Ps_d, n_views, d->views_d, d->n_u, d->n_v, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // Ps is the caller's pageable memory
    d->n_views = n_views;
    d->P_first.assign(Ps, Ps + 12);
    return ECC_OK;
}

ECC_EXPORT int ecc_direct_set_params(ecc_direct* d, double object_radius_mm, double dkappa, int use_fbcc)
{
    if (!d) return fail(ECC_ERR_INVALID_ARGUMENT, "metric is null");
    d->object_radius_mm = object_radius_mm;
    d->dkappa = dkappa;
    d->use_fbcc = use_fbcc;
    return ECC_OK;
}

ECC_EXPORT int ecc_direct_get_object_radius(const ecc_direct* d, double* radius_mm)
{
    if (!d || !radius_mm) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    *radius_mm = direct_radius(d);
    return ECC_OK;
}

ECC_EXPORT int ecc_direct_lines_bound(const ecc_direct* d, int* capacity)
{
    if (!d || !capacity) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    return direct_n_max(d, capacity);
}

ECC_EXPORT int ecc_direct_evaluate(ecc_direct* d, float* cost_nxn, double* cost_sum)
{
    if (!d || !cost_sum) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (d->n_views < 1) return fail(ECC_ERR_INVALID_ARGUMENT, "projection matrices have not been set");
    const int64_t n = d->n_images;  // ref: getNumberOfProjetions() = Is.size()
    if (d->n_views < n) return fail(ECC_ERR_INVALID_ARGUMENT, "fewer projection matrices than images");
    ecc_ctx* ctx = d->ctx;
    int rc = set_device(ctx);
    if (rc) return rc;
    int n_max = 0;
    rc = direct_n_max(d, &n_max);
    if (rc) return rc;
    const int64_t n_pairs = n * (n - 1) / 2;
    HIP_TRY(hipMemsetAsync(d->total_d, 0, sizeof(double), ctx->stream));
    float* cost_d = nullptr;
    if (cost_nxn && n_pairs > 0) {
        if (d->cost_capacity < n * n) {
            if (d->cost_d) {
                HIP_TRY(hipStreamSynchronize(ctx->stream));
                HIP_TRY(hipFree(d->cost_d));
                d->cost_d = nullptr;
            }
            HIP_TRY(hipMalloc((void**)&d->cost_d, sizeof(float) * n * n));
            d->cost_capacity = (int)(n * n);
        }
        cost_d = d->cost_d;
        HIP_TRY(hipMemcpyAsync(cost_d, cost_nxn, sizeof(float) * n * n, hipMemcpyHostToDevice, ctx->stream));
    }
    if (n_pairs > 0) {
        // batches bounded by the grid's y extent and by 256 MB of line integrals in flight
        int64_t batch = (256ll << 20) / (8ll * n_max);
        if (batch < 1) batch = 1;
        if (batch > 65535) batch = 65535;
        if (batch > n_pairs) batch = n_pairs;
        rc = direct_scratch(d, batch, n_max);
        if (rc) return rc;
        for (int64_t first = 0; first < n_pairs; first += batch) {
            EccDirectParams p;
            std::memset(&p, 0, sizeof(p));
            p.images = d->images_d;
            p.imagesT = d->imagesT_d;
            p.image_stride = (int64_t)d->n_u * d->n_v;
            p.views = d->views_d;
            p.pairs = d->pairs_d;
            p.samples = d->samples_d;
            p.pair_metric = d->pair_metric_d;
            p.cost = cost_d;
            p.first = first;
            p.count = std::min<int64_t>(batch, n_pairs - first);
            p.n_views = (int)n;
            p.n_u = d->n_u;
            p.n_v = d->n_v;
            p.n_max = d->n_max_capacity;
            p.object_radius_mm = direct_radius(d);
            p.dkappa = d->dkappa;
            p.use_fbcc = d->use_fbcc ? 1 : 0;
            HIP_TRY(ecc_launch_direct_batch(&p, d->total_d, ctx->stream));
        }
    }
    double total = 0;
    HIP_TRY(hipMemcpyAsync(&total, d->total_d, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (cost_d) HIP_TRY(hipMemcpyAsync(cost_nxn, cost_d, sizeof(float) * n * n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *cost_sum = total;  // ref: ...Direct.cpp:247-259 returns the sum, not the mean
    return ECC_OK;
}

namespace {
int direct_pair_impl(ecc_direct* d, int i, int j, int capacity, int* n_lines, float* rs0, float* rs1, float* kappas,
                     float* lines01, double* metric, const float* kappas_in, int n_kappas_in);
}

ECC_EXPORT int ecc_direct_evaluate_for_image_pair(ecc_direct* d, int i, int j, int capacity, int* n_lines, float* rs0,
                                                  float* rs1, float* kappas, float* lines01, double* metric)
{
    return direct_pair_impl(d, i, j, capacity, n_lines, rs0, rs1, kappas, lines01, metric, nullptr, 0);
}

ECC_EXPORT int ecc_direct_evaluate_for_image_pair_kappas(ecc_direct* d, int i, int j, int n_kappas, const float* kappas_in,
                                                         float* rs0, float* rs1, float* lines01, double* metric)
{
    if (!kappas_in || n_kappas < 1) return fail(ECC_ERR_INVALID_ARGUMENT, "empty kappa grid");
    if (n_kappas > (1 << 24)) return fail(ECC_ERR_INVALID_ARGUMENT, "more than 2^24 epipolar planes");
    int n = 0;
    return direct_pair_impl(d, i, j, n_kappas, &n, rs0, rs1, nullptr, lines01, metric, kappas_in, n_kappas);
}

namespace {
int direct_pair_impl(ecc_direct* d, int i, int j, int capacity, int* n_lines, float* rs0, float* rs1, float* kappas,
                     float* lines01, double* metric, const float* kappas_in, int n_kappas_in)
{
    if (!d || !n_lines) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (d->n_views < 1) return fail(ECC_ERR_INVALID_ARGUMENT, "projection matrices have not been set");
    if (i < 0 || j < 0 || i >= d->n_images || j >= d->n_images || i >= d->n_views || j >= d->n_views)
        return fail(ECC_ERR_INVALID_ARGUMENT, "view index out of range");
    ecc_ctx* ctx = d->ctx;
    int rc = set_device(ctx);
    if (rc) return rc;
    int n_max = 0;
    rc = direct_n_max(d, &n_max);
    if (rc) return rc;
    rc = direct_scratch(d, 1, std::max(n_max, n_kappas_in));
    if (rc) return rc;
    n_max = d->n_max_capacity;
    // debug outputs of pair 0: 6 floats per kappa (lines) + kappa grid + line count + the (i, j) tuple; the grid
    // buffer doubles as the input of a caller-provided grid
    char* dbg = nullptr;
    const size_t lines_b = sizeof(float) * 6 * (size_t)n_max, kap_b = sizeof(float) * (size_t)n_max;
    HIP_TRY(hipMalloc((void**)&dbg, lines_b + 2 * kap_b + 4 * sizeof(int)));
    float* lines_d = reinterpret_cast<float*>(dbg);
    float* kap_d = reinterpret_cast<float*>(dbg + lines_b);
    float* kap_in_d = reinterpret_cast<float*>(dbg + lines_b + kap_b);
    int* count_d = reinterpret_cast<int*>(dbg + lines_b + 2 * kap_b);
    int* idx_d = count_d + 1;
    const int idx[2] = {i, j};
    EccDirectParams p;
    std::memset(&p, 0, sizeof(p));
    p.images = d->images_d;
    p.imagesT = d->imagesT_d;
    p.image_stride = (int64_t)d->n_u * d->n_v;
    p.views = d->views_d;
    p.pairs = d->pairs_d;
    p.idx2 = idx_d;
    p.samples = d->samples_d;
    p.pair_metric = d->pair_metric_d;
    p.pair_lines = count_d;
    p.debug_lines = lines_d;
    p.debug_kappas = kap_d;
    p.first = 0;
    p.count = 1;
    p.n_views = d->n_images;
    p.n_u = d->n_u;
    p.n_v = d->n_v;
    p.n_max = n_max;
    p.object_radius_mm = direct_radius(d);
    p.dkappa = d->dkappa;
    p.use_fbcc = d->use_fbcc ? 1 : 0;
    p.user_kappas = kappas_in ? kap_in_d : nullptr;
    p.n_user_kappas = n_kappas_in;
    std::vector<float> v((size_t)n_max * 2), L((size_t)n_max * 6), K((size_t)n_max);
    int n = 0;
    double m = 0;
    hipError_t e = hipMemcpyAsync(idx_d, idx, sizeof(idx), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(dbg, 0, lines_b + 2 * kap_b + sizeof(int), ctx->stream);
    if (e == hipSuccess && kappas_in)
        e = hipMemcpyAsync(kap_in_d, kappas_in, sizeof(float) * (size_t)n_kappas_in, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = ecc_launch_direct_batch(&p, nullptr, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(v.data(), d->samples_d, sizeof(float) * v.size(), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(L.data(), lines_d, lines_b, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(K.data(), kap_d, kap_b, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&n, count_d, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&m, d->pair_metric_d, sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(dbg);
    HIP_TRY(e);
    *n_lines = n;
    if (n > capacity && (rs0 || rs1 || kappas || lines01))
        return fail(ECC_ERR_INVALID_ARGUMENT, "capacity is smaller than the number of epipolar lines");
    for (int k = 0; k < n; ++k) {
        if (rs0) rs0[k] = v[k];
        if (rs1) rs1[k] = v[(size_t)n_max + k];
        if (kappas) kappas[k] = K[k];
        if (lines01) std::memcpy(lines01 + 6 * (size_t)k, L.data() + 6 * (size_t)k, sizeof(float) * 6);
    }
    if (metric) *metric = m;
    return ECC_OK;
}
}  // namespace

// ---- Metric's free helper functions (host, float64) ------------------------------------------------
ECC_EXPORT void ecc_host_angular_range(const double* P0, const double* P1, double object_radius_mm, double* kappa_first,
                                       double* kappa_second)
{
    double C0[4], C1[4], B[6];
    ecc_host::camera_center(P0, C0);
    ecc_host::camera_center(P1, C1);
    ecc_host::join_points(C0, C1, B);
    const double Pi = 3.14159265358979323846264338327950288419716939937510582;
    // ref: ProjectiveGeometry.hxx:238-268: moment (L3,-L1,L0), direction (-L2,-L4,-L5), distance = |moment|/|direction|
    const double mom = std::sqrt(B[3] * B[3] + B[1] * B[1] + B[0] * B[0]);
    const double dir = std::sqrt(B[2] * B[2] + B[4] * B[4] + B[5] * B[5]);
    const double dist = mom / dir;
    double km = 0.5 * Pi;  // baseline intersects the object: half circle (ref: EpipolarConsistency.cpp:53-55)
    if (!(dist <= object_radius_mm)) km = std::fabs(std::asin(object_radius_mm / dist));
    *kappa_first = -km;
    *kappa_second = km;
}

ECC_EXPORT double ecc_host_angular_step(const double* P0, const double* P1, int n_u, int n_v)
{
    const double r0 = ecc_host::object_radius(P0, n_u, n_v), r1 = ecc_host::object_radius(P1, n_u, n_v);
    double a, b;
    ecc_host_angular_range(P0, P1, r0 > r1 ? r0 : r1, &a, &b);
    return 2.0 * (b - a) / std::sqrt((double)(n_u * n_u + n_v * n_v));
}

ECC_EXPORT void ecc_host_iso_center(const double* Ps, int n_views, double* O)
{
    // A = n I - sum V V^T, b = sum (C - V (V.C)); solve A x = b (3x3, symmetric positive definite for
    // non-parallel rays; the reference solves it with a JacobiSVD, Cramer's rule gives the same x)
    double A[9] = {(double)n_views, 0, 0, 0, (double)n_views, 0, 0, 0, (double)n_views}, b[3] = {0, 0, 0};
    for (int v = 0; v < n_views; ++v) {
        const double* P = Ps + 12 * (size_t)v;
        double C[4];
        ecc_host::camera_center(P, C);
        double V[3] = {P[2], P[5], P[8]};
        const double nv = std::sqrt(V[0] * V[0] + V[1] * V[1] + V[2] * V[2]);
        for (double& x : V) x /= nv;
        const double vc = V[0] * C[0] + V[1] * C[1] + V[2] * C[2];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) A[r + 3 * c] -= V[r] * V[c];
            b[r] += C[r] - V[r] * vc;
        }
    }
    const double det = ecc_host::det3(A, A + 3, A + 6);
    O[0] = ecc_host::det3(b, A + 3, A + 6) / det;
    O[1] = ecc_host::det3(A, b, A + 6) / det;
    O[2] = ecc_host::det3(A, A + 3, b) / det;
    O[3] = 1.0;
}

// ---- cost-balanced shards of the pair range -------------------------------------------------------------------
// Equal-COUNT chunks of the get_ij order are not equal-TIME chunks: the pair kernel's time per pair grows with the
// pair's kappa_max (the sampling curve gets longer, a gather touches more cache lines) and the pairs whose baseline
// passes through the object (kappa_max = pi/2, per-sample path) cost ~5x a short-curve pair; for a circular scan both
// kinds sit in the first rows of the pair triangle.  Measured on MI355X, 400 views of 1024^2 (scripts/shard_step.py): the
// eight equal-count shards of an 8-rank job take 93, 88, 84, 72, 71, 71, 70, 68 us per step.  A least-squares fit over the
// 15 shard timings of 1, 2, 4 and 8 ranks (residual <= 3 us) gives
//     step = 34.7 us + SUM over the shard's pairs of (2.5 ns + 7.0 ns x kappa_max [kappa_max <= pi/4] + 10.4 ns [kappa_max > pi/4]),
// i.e. relative weights 1 + 2.8 kappa_max and 5.2.  ecc_pair_shards_balanced cuts the pair range into contiguous chunks of
// equal model cost (model: 77.5 us for every rank at 8 ranks, 120 us at 4, 206 us at 2).  kappa_max per pair comes from
// the source positions alone (ref: computeK01, EpipolarConsistencyCommon.hxx:115-123,137-145), float64 on the host,
// ~0.3 ms for 79 800 pairs -- once per data set, not per evaluation.
ECC_EXPORT int ecc_pair_shards_balanced(const double* Ps, int n_views, double object_radius_mm, int world, int64_t* bounds)
{
    if (!Ps || !bounds) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (n_views < 2 || world < 1) return fail(ECC_ERR_INVALID_ARGUMENT, "need two views and one rank at least");
    const int64_t n = n_views, n_pairs = n * (n - 1) / 2;
    std::vector<double> C(4 * (size_t)n);
    for (int v = 0; v < n_views; ++v) {
        float c4[4];
        ecc_host::source_position(Ps + 12 * (size_t)v, c4);
        for (int k = 0; k < 4; ++k) C[4 * (size_t)v + k] = c4[k];
    }
    std::vector<double> prefix((size_t)n_pairs + 1);
    prefix[0] = 0.0;
    int64_t q = 0;
    for (int i = 0; i < n_views; ++i)
        for (int j = i + 1; j < n_views; ++j, ++q) {
            const double *a = &C[4 * (size_t)i], *b = &C[4 * (size_t)j];
            const double B01 = a[0] * b[1] - a[1] * b[0], B02 = a[0] * b[2] - a[2] * b[0], B03 = a[0] * b[3] - a[3] * b[0];
            const double B12 = a[1] * b[2] - a[2] * b[1], B13 = a[1] * b[3] - a[3] * b[1], B23 = a[2] * b[3] - a[3] * b[2];
            const double s2 = std::sqrt(B12 * B12 + B02 * B02 + B01 * B01), s3 = std::sqrt(B03 * B03 + B13 * B13 + B23 * B23);
            const double dist = s2 / s3;  // baseline to origin
            double w;
            if (!(dist > object_radius_mm)) w = 5.2;                       // kappa_max = pi/2 (also NaN geometry)
            else {
                const double kmax = std::asin(object_radius_mm / dist);
                w = kmax > 0.78539816339744831 ? 5.2 : 1.0 + 2.8 * kmax;
            }
            prefix[(size_t)q + 1] = prefix[(size_t)q] + w;
        }
    const double total = prefix[(size_t)n_pairs];
    bounds[0] = 0;
    for (int r = 1; r < world; ++r) {
        const double target = total * (double)r / (double)world;
        int64_t b = std::lower_bound(prefix.begin(), prefix.end(), target) - prefix.begin();
        b = std::max<int64_t>(bounds[r - 1], std::min<int64_t>(b, n_pairs));
        bounds[r] = b;
    }
    bounds[world] = n_pairs;
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_balanced_shards(ecc_metric* m, int world, int64_t* bounds)
{
    if (!m || !bounds) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (m->n_views < 2) return fail(ECC_ERR_INVALID_ARGUMENT, "projection matrices have not been set");
    double radius = 0;
    ecc_metric_get_object_radius(m, &radius);
    // the matrices of the last setProjectionMatrices are still in their pinned staging buffer
    return ecc_pair_shards_balanced(m->Ps_h[m->set_generation & 1], m->n_views, radius, world, bounds);
}
