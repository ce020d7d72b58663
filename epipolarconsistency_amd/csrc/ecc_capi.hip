// ecc_capi.hip -- implementation of the C ABI declared in include/ecc_hip.h (host code only).
//
// Host-side flow of the reference that this replaces:
//   RadonIntermediate ctor/compute   ref: LibEpipolarConsistency/RadonIntermediate.cpp:17-31,198-211
//   MetricRadonIntermediate::*        ref: LibEpipolarConsistency/EpipolarConsistencyRadonIntermediate.cpp
// Differences by design: one stream per context and no device-wide syncs between launches; K01 is
// fused into the pair kernel; the mean is reduced on the device in float64 and 8 bytes come back.
// There is NO CPU fallback: without a HIP device every compute entry point fails with
// ECC_ERR_NO_DEVICE / ECC_ERR_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/ecc_hip.h"
#include "ecc_host_geometry.h"
#include "ecc_layout.h"

extern "C" hipError_t ecc_launch_radon(const EccRadonParams* p, int derivative, hipStream_t stream);
extern "C" hipError_t ecc_launch_dtr_border(float* slabs, int64_t slab_stride, int n_img, int n_alpha, int n_t,
                                            int pitch, hipStream_t stream);
extern "C" hipError_t ecc_launch_ramp(float* slabs, int64_t slab_stride, int n_img, int n_alpha, int n_t, int pitch,
                                      const double* h2_d, hipStream_t stream);
extern "C" hipError_t ecc_launch_dtr_import(const float* src, float* slab, int n_alpha, int n_t, int pitch,
                                            hipStream_t stream);
extern "C" hipError_t ecc_launch_dtr_export(const float* slab, float* dst, int n_alpha, int n_t, int pitch,
                                            hipStream_t stream);
extern "C" hipError_t ecc_launch_build_paired(const float* const* slabs_tbl_d, float* paired_d, int64_t paired_stride, int n,
                                              int rows, int pitch, hipStream_t stream);
extern "C" hipError_t ecc_launch_build_quad(const float* const* slabs_tbl_d, float* quads_d, int64_t quad_stride_floats, int n,
                                            int rows, int pitch, hipStream_t stream);
extern "C" hipError_t ecc_launch_k01(const EccPairParams* p, hipStream_t stream);
extern "C" hipError_t ecc_launch_pairs(const EccPairParams* p, hipStream_t stream);
extern "C" int ecc_small_eval_plan(const EccPairParams* p, int* wpp, size_t* lds_bytes);
extern "C" hipError_t ecc_launch_small_eval(const EccPairParams* p, const EccSmallEval* x, hipStream_t stream);
extern "C" hipError_t ecc_launch_preprocess(const EccPreprocessParams* p, hipStream_t stream);
extern "C" size_t ecc_preprocess_lds_bytes(int k);
extern "C" hipError_t ecc_launch_direct_views(const double* Ps_d, int n, EccDirectView* views, int n_u, int n_v,
                                              hipStream_t stream);
extern "C" hipError_t ecc_launch_direct_transpose(const float* src, float* dst, int n, int W, int H, hipStream_t stream);
extern "C" hipError_t ecc_launch_direct_batch(const EccDirectParams* p, double* total, hipStream_t stream);
extern "C" hipError_t ecc_launch_pair_samples(const EccPairSamplesParams* p, hipStream_t stream);
extern "C" hipError_t ecc_launch_sum_pairs(const float* vals, long long count, double* out, void* scratch, hipStream_t stream);
extern "C" hipError_t ecc_launch_sum_pairs_to_host(const float* vals, long long count, double* out, float* values_host, hipStream_t stream);
extern "C" hipError_t ecc_launch_publish_scalar(const double* value_d, double* host_slot_dev, hipStream_t stream);
extern "C" size_t ecc_sum_scratch_bytes();
extern "C" hipError_t ecc_launch_e1(const double* Ps_d, int n, float* PinvTs_d, float* Cs_d, hipStream_t stream);

#define ECC_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            (void)hipGetLastError();                                                              \
            return fail(_e == hipErrorOutOfMemory ? ECC_ERR_OUT_OF_MEMORY : ECC_ERR_HIP,          \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                       \
        }                                                                                         \
    } while (0)

// Device slab shared by the dtrs of one batch; freed when the last handle goes away.
struct Slab {
    float* ptr = nullptr;
    int device = 0;
    ~Slab()
    {
        if (ptr) {
            (void)hipSetDevice(device);
            (void)hipFree(ptr);
        }
    }
};

}  // namespace

struct ecc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool timing = false;
    int radon_arithmetic = ECC_RADON_EXACT;  // ecc_radon_set_arithmetic
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // pair, radon, preprocess start/stop
    bool ev_valid[3] = {false, false, false};
    // trig table cache for the Radon kernel
    float* trig_d = nullptr;
    int trig_n_alpha = 0;
    // circular-convolution kernel of Filter::Ramp (2*n_t doubles), cached per n_t
    double* ramp_d = nullptr;
    int ramp_n_t = 0;
    // constant tables of the pair kernel's polynomial path
    EccPolyTables* poly_d = nullptr;
    // arena of ecc_preprocess, reused between calls and freed with the context: device tables + their pinned host
    // image (uploaded asynchronously; pre_ev marks the last upload, so the host image is not rewritten under a copy
    // in flight) and the scratch stack of the host-input / in-place forms
    char* pre_tables_d = nullptr;
    char* pre_tables_h = nullptr;
    size_t pre_tables_cap = 0;
    hipEvent_t pre_ev = nullptr;
    bool pre_ev_recorded = false;
    float* pre_scratch_d[2] = {nullptr, nullptr};
    size_t pre_scratch_cap[2] = {0, 0};
    // transposed copy of (a sub-batch of) the projection images for the Radon kernel's transposed tiles
    float* radon_T_d = nullptr;
    size_t radon_T_cap = 0;  // floats
    // one slab of scratch for ecc_radon_compute_linear
    float* linear_scratch_d = nullptr;
    size_t linear_scratch_cap = 0;  // floats
};

struct ecc_dtr {
    ecc_ctx* ctx = nullptr;
    std::shared_ptr<Slab> owner;  // null when wrapping caller memory
    float* base = nullptr;
    int n_alpha = 0, n_t = 0, n_u = 0, n_v = 0, filter = 0, pitch = 0;
};

struct ecc_metric {
    ecc_ctx* ctx = nullptr;
    std::vector<ecc_dtr*> dtrs;
    int n_alpha = 0, n_t = 0, n_u = 0, n_v = 0, pitch = 0;
    bool is_derivative = true;
    float step_alpha = 0, step_t = 0;
    // parameters
    double object_radius_mm = 0, dkappa = 0;
    int use_corr = 0;
    int sampling = ECC_SAMPLING_AUTO;  // ecc_metric_set_sampling
    // projections
    int n_views = 0;
    std::vector<double> P_first;  // first projection matrix (object radius estimate)
    // device state
    const float** dtr_table_d = nullptr;     // the dtrs' slabs (borrowed)
    float* paired_d = nullptr;               // row-paired copies of all dtrs (owned; what the pair kernel samples)
    const float** paired_table_d = nullptr;  // per dtr: base of its paired copy
    float* quads_d = nullptr;                // row-quad copies of all dtrs (owned; sampled by the pairs with kappa_max > pi/4), or null
    const float** quads_table_d = nullptr;
    int64_t quad_floats = 0;                 // floats per row-quad copy
    float* Cs_d = nullptr;
    float* PinvTs_d = nullptr;
    int geom_capacity = 0;
    float* pair_values_d = nullptr;
    int64_t pair_capacity = 0;
    float* cost_d = nullptr;
    int cost_capacity = 0;
    int32_t* indices_d = nullptr;
    int64_t indices_capacity = 0;
    float* K01_d = nullptr;
    int64_t K01_capacity = 0;
    EccPairRecord* records_d = nullptr;  // per-pair geometry between k01_kernel and pairs_kernel
    int64_t records_capacity = 0;
    double* sum_d = nullptr;
    void* sum_scratch_d = nullptr;  // partials + ticket of the multi-workgroup sum (zeroed; pairs_kernel.hip)
    double* Ps_d = nullptr;   // n x 12 float64 as handed over by the caller
    // pinned host staging, mapped into the device's address space (zero-copy: the 38 KB of matrices and the 8-byte
    // result cross PCIe inside the kernels, no copy commands).  Two matrix buffers, used alternately: an evaluate
    // call returns only after the stream has executed everything up to its result, so the buffer of the call before
    // the previous one is free without asking the stream (set_generation / done_generation below).
    double* Ps_h[2] = {nullptr, nullptr};
    double* Ps_h_dev[2] = {nullptr, nullptr};
    uint64_t set_generation = 0;   // number of e1 launches so far; launch g reads Ps_h[g & 1]
    uint64_t done_generation = 0;  // every e1 launch up to this one is known to have completed
    double* sum_h = nullptr;       // 64-byte slot; [0] = the result, written by sum_pairs_kernel with a system-scope store
    double* sum_h_dev = nullptr;
    // ecc_metric_set_incremental: the pair values of the last evaluation of one pair range, the matrices and parameters
    // they belong to, and a pinned, device-mapped list buffer (4 indices + 1 slot per re-evaluated pair)
    int incremental = 0;
    bool cache_valid = false;
    int64_t cache_first = 0, cache_count = 0;
    int cache_n_views = 0, cache_use_corr = 0, cache_sampling = 0;
    double cache_radius = 0, cache_dkappa = 0;
    std::vector<double> cache_Ps;
    float* cache_values_d = nullptr;
    int64_t cache_capacity = 0;
    int32_t* list_h = nullptr;
    int32_t* list_h_dev = nullptr;
    int64_t list_capacity = 0;  // pairs
    std::vector<int> scratch_changed;        // reused between evaluations (no heap traffic on the optimiser's path)
    std::vector<char> scratch_is_changed;
    std::vector<int32_t> scratch_idx, scratch_slots;
    int64_t last_evaluated_pairs = 0;
    // ecc_metric_set_record_reuse: the per-pair records (k01_kernel's output) of the last all-pairs / range evaluation
    // stay in records_d together with the matrices and parameters they belong to; the next evaluation of the same range
    // refits only the pairs with a changed matrix.  E1 is deferred to the evaluation for the same reason.
    int record_reuse = 1;
    bool e1_pending = false;   // matrices were staged since PinvTs_d / Cs_d were last known to be current (ensure_e1 finds out)
    // the matrices PinvTs_d / Cs_d on the device were made from, view by view (e1_kernel: all views; the patch lists of the
    // reuse path and of the one-launch path: the listed views)
    std::vector<double> dev_Ps;
    bool dev_valid = false;
    bool eager_e1 = false;     // the last range was too small for record reuse: set_projections launches e1_kernel itself
    bool rec_valid = false;
    int64_t rec_first = 0, rec_count = 0;
    int rec_n_views = 0, rec_mode = 0;
    float rec_radius = 0, rec_dkappa = 0, rec_tol = 0;
    std::vector<double> rec_Ps;
    // pinned, device-mapped lists of the reuse path, two used alternately: per pair 4 indices + slot + 2 patch refs,
    // per changed view 16 floats + its index
    int32_t* reuse_h[2] = {nullptr, nullptr};
    int32_t* reuse_h_dev[2] = {nullptr, nullptr};
    int64_t reuse_words[2] = {0, 0};
    hipEvent_t reuse_ev[2] = {nullptr, nullptr};  // recorded after the k01 launch that read list b (asynchronous callers)
    bool reuse_ev_used[2] = {false, false};
    uint64_t reuse_gen = 0;
    std::vector<int> scratch_patched;
    std::vector<int32_t> scratch_refs;
    std::vector<int32_t> scratch_patch_of;
    // second stream of the reuse path: refit + list launch of the changed pairs run there while the all-pairs launch
    // (which skips them) already runs on the context's stream
    hipStream_t side_stream = nullptr;
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    // ecc_metric_set_small_eval: evaluations of at most ECC_SMALL_EVAL_MAX_PAIRS pairs as ONE launch (small_eval_kernel.hip).
    // E1 of the views whose matrix changed since the device arrays were made is done on the host and handed over in the
    // kernel arguments (dev_Ps above says which views those are).
    int small_eval = 1;
    int32_t* sidx_h = nullptr;    // index list of a fused index-list evaluation (pinned, device-mapped)
    int32_t* sidx_h_dev = nullptr;
    int64_t sidx_capacity = 0;    // pairs
    float* svals_h = nullptr;     // pair values a caller wants on the host (pinned, device-mapped)
    float* svals_h_dev = nullptr;
    int64_t svals_capacity = 0;
    unsigned* small_ticket_d = nullptr;
    int64_t small_pending_count = 0;  // > 0: the result slot will receive the "done" word of a one-launch evaluation of that many pairs
};

namespace {

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

bool result_polling_enabled()
{
    static const bool on = [] {
        const char* e = std::getenv("ECC_RESULT_WAIT");  // "stream": wait for the stream instead (A/B measurements)
        return !(e && std::strcmp(e, "stream") == 0);
    }();
    return on;
}

void arm_result(ecc_metric* m)
{
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

template <class T>
int ensure_capacity(T** ptr, int64_t* cap, int64_t need, hipStream_t stream)
{
    if (*cap >= need && *ptr) return ECC_OK;
    if (*ptr) {
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipFree(*ptr));
        *ptr = nullptr;
        *cap = 0;
    }
    HIP_TRY(hipMalloc((void**)ptr, (size_t)need * sizeof(T)));
    *cap = need;
    return ECC_OK;
}

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
int ecc_set_error(int code, const char* msg) { return fail(code, msg); }
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
        const char* e = std::getenv("ECC_RECORD_REUSE");  // 0: off, 1: default, 2: for every size
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
    // row-quad copies: opt-in (ECC_QUAD_COPIES=1; 4x the slab memory, see pairs_kernel.hip); offsets must fit 32 bits
    std::vector<const float*> qtable(n_dtrs);
    m->quad_floats = (int64_t)((m->n_alpha + 1 + 3) / 4) * m->pitch * 16;
    const char* quads_env = std::getenv("ECC_QUAD_COPIES");
    const bool want_quads = m->quad_floats * 4 < ((int64_t)1 << 32) && quads_env && quads_env[0] == '1';
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

namespace {
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
}  // namespace

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
    if (m->eager_e1) return ensure_e1(m);  // the last evaluation needed it on the device and skipped nothing: launch it now
    return ECC_OK;
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
    else *radius_mm = ecc_host::object_radius(m->P_first.data(), m->n_u, m->n_v);
    return ECC_OK;
}

namespace {

// ECC_SAMPLING_AUTO -> the mode one evaluation of `count` pairs runs in (include/ecc_hip.h)
int resolve_sampling(const ecc_metric* m, int64_t count)
{
    if (m->sampling != ECC_SAMPLING_AUTO) return m->sampling;
    return count <= ECC_SAMPLING_AUTO_REFERENCE_PAIRS ? ECC_SAMPLING_REFERENCE : ECC_SAMPLING_POLYNOMIAL;
}

// mode_count: the size of the EVALUATION the launch belongs to -- n (n - 1) / 2 for all-pairs evaluations and every shard
// (range) of one, the list length for index lists -- which is what ECC_SAMPLING_AUTO resolves from: a shard of an
// evaluation runs in the mode of the whole, whatever its own size, so the sum of G shard sums is the one-device sum's
// arithmetic and a re-balanced shard does not change mode.
int fill_pair_params(ecc_metric* m, EccPairParams* p, int64_t mode_count, bool need_e1 = true)
{
    if (m->n_views < 1) return fail(ECC_ERR_INVALID_ARGUMENT, "projection matrices have not been set");
    if (need_e1) {
        const int rc1 = ensure_e1(m);
        if (rc1) return rc1;
    }
    const int64_t count = mode_count;
    double radius = 0;
    ecc_metric_get_object_radius(m, &radius);
    std::memset(p, 0, sizeof(*p));
    p->dtrs = m->paired_table_d;  // the pair kernel samples the row-paired copies
    p->Cs = m->Cs_d;
    p->PinvTs = m->PinvTs_d;
    p->n_views = m->n_views;
    p->n_alpha = m->n_alpha;
    p->n_t = m->n_t;
    p->pitch = m->pitch;
    // launcher arguments, ref: ...RadonIntermediate.cu:320-358 (fp32, same expressions)
    p->n_x2 = m->n_u * 0.5f;
    p->n_y2 = m->n_v * 0.5f;
    p->object_radius_mm = (float)radius;
    const float image_diagonal = m->n_t * m->step_t * 2.f;
    p->num_samples = image_diagonal;
    p->range_t = m->n_t * m->step_t;
    p->dkappa_user = (float)m->dkappa;
    const float Pi = 3.14159265359f;
    int max_num_samples = p->dkappa_user <= 0.0f ? (int)image_diagonal : (int)(Pi * 0.5f / p->dkappa_user);
    p->k_limit = (max_num_samples + 255) / 256 * 256;
    p->is_derivative = m->is_derivative ? 1 : 0;
    p->use_corr = m->use_corr ? 1 : 0;
    int rc = ensure_poly_tables(m->ctx);
    if (rc) return rc;
    // without the tables k01_kernel fits nothing and marks every pair for the per-sample path
    p->poly = resolve_sampling(m, count) == ECC_SAMPLING_POLYNOMIAL ? m->ctx->poly_d : nullptr;
    p->slabs = m->dtr_table_d;  // ECC_SAMPLING_REFERENCE samples the dtrs themselves (clamped taps), not the paired copies
    p->reference_arithmetic = resolve_sampling(m, count) == ECC_SAMPLING_REFERENCE ? 1 : 0;
    // few pairs: all four waves of a workgroup on one pair (a function of the FULL range's size, like the mode itself)
    p->reference_split = (p->reference_arithmetic && count <= 2048) ? 4 : 1;
    p->wide_offsets = ((int64_t)(m->n_alpha + 1) * m->pitch * 8 >= (int64_t)1 << 24) ? 1 : 0;
    p->quads = m->quads_table_d;
    p->quad_group_bytes = (unsigned)m->pitch * 64u;
    {
        static const float tol = [] {
            const char* e = std::getenv("ECC_POLY_TOL");  // experiments only
            return e ? (float)std::atof(e) : 2e-8f;
        }();
        p->economise_tol = tol;
    }
    return ECC_OK;
}

// The pinned list buffer b of the reuse path with room for `words` 32-bit words.
int ensure_reuse_list(ecc_metric* m, int b, int64_t words)
{
    if (m->reuse_words[b] >= words) return ECC_OK;
    HIP_TRY(wait_stream_spin(m->ctx->stream));  // a launch may still be reading the old buffer
    if (m->reuse_h[b]) HIP_TRY(hipHostFree(m->reuse_h[b]));
    m->reuse_h[b] = nullptr;
    m->reuse_words[b] = 0;
    const int64_t cap = std::max<int64_t>(2 * words, 16384);
    HIP_TRY(hipHostMalloc((void**)&m->reuse_h[b], sizeof(int32_t) * cap, hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void**)&m->reuse_h_dev[b], m->reuse_h[b], 0));
    m->reuse_words[b] = cap;
    return ECC_OK;
}

// One launch for an evaluation of at most ECC_SMALL_EVAL_MAX_PAIRS pairs (small_eval_kernel.hip; ref for what it replaces:
// ...RadonIntermediate.cu:300-409, two kernels and two device-wide syncs).  p: the launch as fill_pair_params and the caller
// left it (first, count, pair_values, cost; indices ignored -- the list comes as idx4_host).  E1 of the views whose
// matrix differs from what PinvTs_d / Cs_d were made from (dev_Ps) is computed here on the host with the code e1_kernel
// compiles (ecc_host_geometry.h, bit-identical) and travels in the kernel arguments -- at most ECC_SMALL_PATCH_MAX views;
// beyond that (the first call, a new trajectory) e1_kernel runs in front.  The kept records are not touched.
// *taken = false: the evaluation does not qualify and nothing was launched.
unsigned long long* g_small_dbg = nullptr;

// Pinned, device-mapped staging of index-list evaluations: the list (4 ints per pair) and the pair values.
int ensure_small_host_buffers(ecc_metric* m, int64_t idx_pairs, int64_t value_pairs)
{
    if (m->sidx_capacity < idx_pairs) {
        HIP_TRY(wait_stream_spin(m->ctx->stream));  // nothing may still be reading the old buffer
        if (m->sidx_h) HIP_TRY(hipHostFree(m->sidx_h));
        m->sidx_h = nullptr;
        m->sidx_capacity = 0;
        const int64_t cap = std::max<int64_t>(2 * idx_pairs, 1024);
        HIP_TRY(hipHostMalloc((void**)&m->sidx_h, sizeof(int32_t) * 4 * cap, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void**)&m->sidx_h_dev, m->sidx_h, 0));
        m->sidx_capacity = cap;
    }
    if (m->svals_capacity < value_pairs) {
        HIP_TRY(wait_stream_spin(m->ctx->stream));
        if (m->svals_h) HIP_TRY(hipHostFree(m->svals_h));
        m->svals_h = nullptr;
        m->svals_capacity = 0;
        const int64_t cap = std::max<int64_t>(2 * value_pairs, 1024);
        HIP_TRY(hipHostMalloc((void**)&m->svals_h, sizeof(float) * cap, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void**)&m->svals_h_dev, m->svals_h, 0));
        m->svals_capacity = cap;
    }
    return ECC_OK;
}
constexpr unsigned long long ECC_SMALL_DONE_TOKEN = 0x7ff8ecc0d04e0001ull;  // a NaN payload: not a sum, not ECC_RESULT_PENDING

// The float64 sum of `count` <= 4096 pair values exactly as sum_pairs_kernel forms it (pairs_kernel.hip; ref:
// ...RadonIntermediate.cpp:216-224): thread t of its 1024 holds ((0 + v[4t]) + (0 + v[4t+1])) + ((0 + v[4t+2]) + (0 + v[4t+3]))
// (one float4 at most for such a count), thread 0 then adds the up to three values past the last float4, the 64 threads
// of a wave are combined by the shuffle-down tree (offsets 32 ... 1), the 16 wave sums are added in order.  IEEE binary64
// additions in the same order: the same bits.
double small_sum_on_host(const float* v, int64_t count)
{
    const int64_t n4 = count >> 2;
    double tot = 0.0;
    for (int w = 0; w < 16; ++w) {
        double a[64];
        for (int l = 0; l < 64; ++l) {
            const int64_t t = 64 * w + l;
            double acc = 0.0;
            if (t < n4) {
                const double a0 = 0.0 + (double)v[4 * t], a1 = 0.0 + (double)v[4 * t + 1], a2 = 0.0 + (double)v[4 * t + 2],
                             a3 = 0.0 + (double)v[4 * t + 3];
                acc = (a0 + a1) + (a2 + a3);
            }
            if (t == 0)
                for (int64_t k = n4 << 2; k < count; ++k) acc += (double)v[k];
            a[l] = acc;
        }
        for (int off = 32; off > 0; off >>= 1)
            for (int l = 0; l < off; ++l) a[l] += a[l + off];  // what lane 0 of __shfl_down's tree ends up with
        tot += a[0];
    }
    return tot;
}

// Waits for the "done" word of the one-launch evaluation (the result slot, armed by the caller) and adds the values.
hipError_t wait_small_eval(ecc_metric* m, int64_t count, double* sum)
{
    double token = 0.0;
    const hipError_t e = wait_result(m, m->ctx->stream, &token);
    if (e != hipSuccess) return e;
    unsigned long long bits;
    std::memcpy(&bits, &token, sizeof(bits));
    if (bits != ECC_SMALL_DONE_TOKEN) return hipErrorUnknown;  // the kernel found its argument views inconsistent
    std::atomic_thread_fence(std::memory_order_acquire);
    *sum = small_sum_on_host(m->svals_h, count);
    return hipSuccess;
}

// What the synchronous evaluate calls wait for: the sum kernel's store, or the hand-over of a one-launch evaluation.
hipError_t wait_sum(ecc_metric* m, double* sum)
{
    if (m->small_pending_count > 0) {
        const int64_t count = m->small_pending_count;
        m->small_pending_count = 0;
        return wait_small_eval(m, count, sum);
    }
    return wait_result(m, m->ctx->stream, sum);
}
int try_small_eval(ecc_metric* m, EccPairParams p, const int32_t* idx4_host, bool* taken)
{
    *taken = false;
    int wpp = 0;
    size_t lds = 0;
    if (!m->small_eval || !ecc_small_eval_plan(&p, &wpp, &lds)) return ECC_OK;

    ecc_ctx* ctx = m->ctx;
    const int n = m->n_views;
    if (!m->small_ticket_d) {
        HIP_TRY(hipMalloc((void**)&m->small_ticket_d, sizeof(unsigned)));
        HIP_TRY(hipMemsetAsync(m->small_ticket_d, 0, sizeof(unsigned), ctx->stream));
    }
    EccSmallEval x;
    std::memset(&x, 0, sizeof(x));
    const int slot = (int)(m->set_generation & 1);
    const double* Pcur = m->Ps_h[slot];
    // views whose geometry on the device is behind the current matrices
    std::vector<int>& stale = m->scratch_changed;
    stale.clear();
    const bool dev_known = m->dev_valid && (int64_t)m->dev_Ps.size() == 12 * (int64_t)n;
    if (dev_known && m->e1_pending)
        for (int v = 0; v < n && (int)stale.size() <= ECC_SMALL_PATCH_MAX; ++v)
            if (std::memcmp(Pcur + 12 * v, m->dev_Ps.data() + 12 * v, sizeof(double) * 12) != 0) stale.push_back(v);
    if (!dev_known || (int)stale.size() > ECC_SMALL_PATCH_MAX) {
        const int rc = ensure_e1(m);  // the first call, a new trajectory: e1_kernel, ordered before the launch below
        if (rc) return rc;
    } else {
        for (size_t e = 0; e < stale.size(); ++e) {  // ref: ...RadonIntermediate.cpp:134-163
            const int v = stale[e];
            ecc_host::pinv_transpose(Pcur + 12 * v, x.patch_geo[e]);
            ecc_host::source_position(Pcur + 12 * v, x.patch_geo[e] + 12);
            x.patch_views[e] = v;
            std::memcpy(m->dev_Ps.data() + 12 * v, Pcur + 12 * v, sizeof(double) * 12);  // workgroup 0 stores the entry
        }
        x.patch_count = (int)stale.size();
        m->e1_pending = false;
    }
    p.PinvTs = m->PinvTs_d;
    p.Cs = m->Cs_d;
    {
        const int rcb = ensure_small_host_buffers(m, idx4_host ? p.count : 0, p.count);
        if (rcb) return rcb;
    }
    p.indices = nullptr;
    if (idx4_host) {
        std::memcpy(m->sidx_h, idx4_host, sizeof(int32_t) * 4 * (size_t)p.count);
        p.indices = m->sidx_h_dev;
    }
    x.ticket = m->small_ticket_d;
    x.values_host = m->svals_h_dev;
    // the "done" word: the metric's pinned result slot, armed by the caller; the token is never a value a sum kernel stores
    x.done_out = reinterpret_cast<unsigned long long*>(m->sum_h_dev);
    x.done_token = ECC_SMALL_DONE_TOKEN;
    static unsigned long long* dbg_d = [] {  // experiments only: ECC_SMALL_DEBUG=1, read back by ecc_debug_small_stamps
        unsigned long long* d = nullptr;
        if (std::getenv("ECC_SMALL_DEBUG")) (void)hipMalloc((void**)&d, sizeof(unsigned long long) * 4 * 4096);
        return d;
    }();
    x.dbg = dbg_d;
    g_small_dbg = dbg_d;

    std::atomic_thread_fence(std::memory_order_seq_cst);  // the host's writes to pinned memory before the doorbell
    if (ctx->timing) HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
    HIP_TRY(ecc_launch_small_eval(&p, &x, ctx->stream));
    if (ctx->timing) {
        HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
        ctx->ev_valid[0] = true;
    }
    m->eager_e1 = false;  // the views that changed are patched by the next launch: ecc_metric_set_projections does not launch E1
    m->last_evaluated_pairs = p.count;
    m->small_pending_count = p.count;
    *taken = true;
    return ECC_OK;
}

// k01_kernel + pairs_kernel (+ sum) over the pair range [first, first + count).
// Record reuse (default on, ecc_metric_set_record_reuse): a pair's record is a function of its two matrices and the
// parameters only.  When this range was evaluated before with the same parameters and at most a quarter of the matrices
// differ from the ones its records were made from, k01_kernel runs over an index list of the pairs that contain a
// changed view (8 lanes per fit up to 4096 pairs) and writes each record into its slot of the kept array; E1 of the
// changed views is done on the host with the device's own code (ecc_host_geometry.h, bit-identical) and reaches the
// kernel through pinned memory, so e1_kernel is not launched at all.  pairs_kernel then samples EVERY pair as always:
// the evaluation's result is bit-identical to one that refits everything (tests/test_gpu_record_reuse.py).
// synchronous: the caller waits for the result before it returns (the list buffers need no event then).
int launch_range(ecc_metric* m, int64_t first, int64_t count, float* pair_values_d, float* cost_d, float* K01_d,
                 double* sum_d, bool synchronous = false)
{
    ecc_ctx* ctx = m->ctx;
    const int64_t n = m->n_views;
    const int64_t n_pairs = n * (n - 1) / 2;
    if ((int)m->dtrs.size() < m->n_views)
        return fail(ECC_ERR_INVALID_ARGUMENT, "fewer Radon intermediates than projection matrices");
    if (first < 0 || count < 0 || first + count > n_pairs)
        return fail(ECC_ERR_INVALID_ARGUMENT, "pair range outside [0, n(n-1)/2)");
    EccPairParams p;
    int rc = fill_pair_params(m, &p, n_pairs, /*need_e1=*/false);
    if (rc) return rc;
    rc = ensure_capacity(&m->records_d, &m->records_capacity, count > 0 ? count : 1, ctx->stream);
    if (rc) return rc;
    p.first = first;
    p.count = count;
    p.pair_values = pair_values_d;
    p.cost = cost_d;
    p.K01_out = K01_d;
    p.records = m->records_d;
    m->last_evaluated_pairs = count;
    if (sum_d && sum_d == m->sum_h_dev && pair_values_d) {
        // few pairs, a caller that waits for the result: ONE launch (small_eval_kernel.hip); the kept records are not touched.
        // The result slot then receives the "done" word and the caller's wait_sum adds the values on the host.
        bool taken = false;
        rc = try_small_eval(m, p, nullptr, &taken);
        if (rc) return rc;
        if (taken) return ECC_OK;
    }

    const int mode = p.reference_arithmetic ? ECC_SAMPLING_REFERENCE : (p.poly ? ECC_SAMPLING_POLYNOMIAL : ECC_SAMPLING_PER_SAMPLE);
    const double* Pcur = m->Ps_h[m->set_generation & 1];
    bool reused = false;
    // Small ranges gain nothing: up to ECC_RECORD_REUSE_MIN_PAIRS pairs the refit of everything is one 7-us launch of
    // k01_kernel<8>, and a short pair kernel cannot hide the list launches of the two-stream form behind it -- the moved
    // view's own pairs include its neighbours', whose waves run 40-50 us (64 views, 2016 pairs: 61 us per step with two
    // streams against 40 us refitting everything).  Mode 2 (tests) applies the two-stream form at every size.
    const bool size_ok = m->record_reuse >= 2 || count > ECC_RECORD_REUSE_MIN_PAIRS;
    m->eager_e1 = !m->record_reuse || !size_ok;
    const bool rec_match = m->record_reuse && size_ok && m->rec_valid && !K01_d && count > 0 && m->rec_first == first && m->rec_count == count &&
                           m->rec_n_views == (int)n && m->rec_mode == mode && m->rec_radius == p.object_radius_mm &&
                           m->rec_dkappa == p.dkappa_user && m->rec_tol == p.economise_tol && (int64_t)m->rec_Ps.size() == 12 * n;
    m->rec_valid = false;  // until everything below is enqueued
    bool pairs_launched = false;
    if (rec_match) {
        std::vector<int>& changed = m->scratch_changed;
        changed.clear();
        for (int64_t v = 0; v < n; ++v)
            if (std::memcmp(Pcur + 12 * v, m->rec_Ps.data() + 12 * v, sizeof(double) * 12) != 0) changed.push_back((int)v);
        // views whose geometry on the device is not that of the current matrices although their records are (an E1 launch
        // or a patch list of another call in between): they need a patch entry too, but no refit
        std::vector<int>& patched = m->scratch_patched;
        patched = changed;
        const bool dev_known = m->dev_valid && (int64_t)m->dev_Ps.size() == 12 * n;
        if (dev_known) {
            for (int64_t v = 0; v < n; ++v)
                if (std::memcmp(Pcur + 12 * v, m->dev_Ps.data() + 12 * v, sizeof(double) * 12) != 0 &&
                    std::memcmp(Pcur + 12 * v, m->rec_Ps.data() + 12 * v, sizeof(double) * 12) == 0)
                    patched.push_back((int)v);
        }
        if (dev_known && (int64_t)patched.size() * 4 <= n) {
            const int64_t C = (int64_t)changed.size(), Cp = (int64_t)patched.size();
            // Two streams: the all-pairs launch that SKIPS the pairs of the changed views starts at once on the context's
            // stream; the refit of those pairs and their own list launch follow on the metric's side stream, hidden behind
            // it; the sum waits for both.  (Not with a cost image -- the list launch does not write it --, not in the
            // reference arithmetic -- evaluations of at most 512 pairs --, not beyond 512 views: the skip set is a
            // 512-bit kernel argument.)
            bool split = C > 0 && !cost_d && !p.reference_arithmetic && n <= 32 * ECC_SKIP_WORDS &&
                         (m->record_reuse >= 2 || count >= ECC_RECORD_REUSE_SPLIT_PAIRS);
            if (split && (!m->side_stream || !m->fork_ev || !m->join_ev)) {  // all three or none (advisor, round 3)
                if (hipStreamCreateWithFlags(&m->side_stream, hipStreamNonBlocking) != hipSuccess ||
                    hipEventCreateWithFlags(&m->fork_ev, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&m->join_ev, hipEventDisableTiming) != hipSuccess) {
                    (void)hipGetLastError();
                    if (m->side_stream) (void)hipStreamDestroy(m->side_stream);
                    if (m->fork_ev) (void)hipEventDestroy(m->fork_ev);
                    if (m->join_ev) (void)hipEventDestroy(m->join_ev);
                    m->side_stream = nullptr;
                    m->fork_ev = m->join_ev = nullptr;
                    split = false;
                }
            }
            if (split) {
                // whatever the caller queued on the context's stream before this call comes first for the side stream too
                HIP_TRY(hipEventRecord(m->fork_ev, ctx->stream));
                EccPairParams pa = p;
                pa.skip_enabled = 1;
                for (int v : changed) pa.skip_mask[v >> 5] |= 1u << (v & 31);
                if (ctx->timing) HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
                HIP_TRY(ecc_launch_pairs(&pa, ctx->stream));
                if (ctx->timing) {
                    HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
                    ctx->ev_valid[0] = true;
                }
                pairs_launched = true;
            }
            std::vector<char>& is_changed = m->scratch_is_changed;
            std::vector<int32_t>&idx = m->scratch_idx, &slots = m->scratch_slots, &refs = m->scratch_refs, &patch_of = m->scratch_patch_of;
            is_changed.assign((size_t)n, 0);
            patch_of.assign((size_t)n, -1);
            for (int v : changed) is_changed[v] = 1;
            for (size_t e = 0; e < patched.size(); ++e) patch_of[patched[e]] = (int32_t)e;
            idx.clear();
            slots.clear();
            refs.clear();
            for (int v : changed)
                for (int64_t u = 0; u < n; ++u) {
                    if (u == v || (is_changed[u] && u < v)) continue;  // a pair of two changed views once
                    const int64_t i = u < v ? u : v, j = u < v ? v : u;
                    const int64_t ij = i * n - i * (i + 1) / 2 + (j - i - 1);  // get_ij order
                    if (ij < first || ij >= first + count) continue;
                    idx.insert(idx.end(), {(int32_t)i, (int32_t)j, (int32_t)i, (int32_t)j});
                    slots.push_back((int32_t)(ij - first));
                    refs.push_back(patch_of[i]);
                    refs.push_back(patch_of[j]);
                }
            const int64_t L = (int64_t)slots.size();
            const int b = (int)(m->reuse_gen++ & 1);
            // L = 0 (no pair of this range contains a changed view): nothing to refit and nothing launched; rec_Ps keeps
            // the old matrices of those views, which is what PinvTs / Cs on the device still correspond to
            if (L > 0) {
                rc = ensure_reuse_list(m, b, 7 * L + 17 * Cp);
                if (rc) return rc;
                if (m->reuse_ev_used[b]) {  // an asynchronous caller: the launches that read this buffer two calls ago
                    HIP_TRY(hipEventSynchronize(m->reuse_ev[b]));
                    m->reuse_ev_used[b] = false;
                }
                int32_t* h = m->reuse_h[b];
                std::memcpy(h, idx.data(), sizeof(int32_t) * 4 * L);
                std::memcpy(h + 4 * L, slots.data(), sizeof(int32_t) * L);
                std::memcpy(h + 5 * L, refs.data(), sizeof(int32_t) * 2 * L);
                float* geo = reinterpret_cast<float*>(h + 7 * L);
                int32_t* views = h + 7 * L + 16 * Cp;
                for (int64_t e = 0; e < Cp; ++e) {  // E1 of the patched views (ref: ...RadonIntermediate.cpp:134-163)
                    ecc_host::pinv_transpose(Pcur + 12 * patched[e], geo + 16 * e);
                    ecc_host::source_position(Pcur + 12 * patched[e], geo + 16 * e + 12);
                    views[e] = patched[e];
                }
                EccPairParams q = p;  // k01_kernel over the list
                q.indices = m->reuse_h_dev[b];
                q.record_slots = m->reuse_h_dev[b] + 4 * L;
                q.patch_ref = m->reuse_h_dev[b] + 5 * L;
                q.patch_geo = reinterpret_cast<const float*>(m->reuse_h_dev[b] + 7 * L);
                q.patch_views = m->reuse_h_dev[b] + 7 * L + 16 * Cp;
                q.patch_count = (int)Cp;
                q.first = 0;
                q.count = L;
                q.cost = nullptr;
                q.pair_values = nullptr;
                hipStream_t ks = split ? m->side_stream : ctx->stream;
                // from here on an early return must not leave side-stream work un-joined (the list buffers are reused
                // by later calls, which wait on the context's stream only): SIDE_TRY drains the side stream first
#define SIDE_TRY(expr)                                                        \
    do {                                                                      \
        const hipError_t _s = (expr);                                         \
        if (_s != hipSuccess) {                                               \
            if (split) (void)hipStreamSynchronize(m->side_stream);            \
            HIP_TRY(_s);                                                      \
        }                                                                     \
    } while (0)
                if (split) SIDE_TRY(hipStreamWaitEvent(m->side_stream, m->fork_ev, 0));
                SIDE_TRY(ecc_launch_k01(&q, ks));
                if (split) {  // the changed pairs' own launch: records and values in their slots
                    q.pair_values = pair_values_d;
                    q.value_slots = q.record_slots;
                    SIDE_TRY(ecc_launch_pairs(&q, m->side_stream));
                    SIDE_TRY(hipEventRecord(m->join_ev, m->side_stream));
                    SIDE_TRY(hipStreamWaitEvent(ctx->stream, m->join_ev, 0));
                }
#undef SIDE_TRY
                if (!synchronous) {
                    if (!m->reuse_ev[b]) HIP_TRY(hipEventCreateWithFlags(&m->reuse_ev[b], hipEventDisableTiming));
                    HIP_TRY(hipEventRecord(m->reuse_ev[b], ks));
                    m->reuse_ev_used[b] = true;
                }
                for (int v : changed) std::memcpy(m->rec_Ps.data() + 12 * v, Pcur + 12 * v, sizeof(double) * 12);
                for (int v : patched) std::memcpy(m->dev_Ps.data() + 12 * v, Pcur + 12 * v, sizeof(double) * 12);
                m->e1_pending = false;  // workgroup 0 of the list launch stores the patches: PinvTs / Cs are current again
            }
            // (L = 0: nothing was launched; dev_Ps says which views of the device arrays are behind, ensure_e1 will look)
            reused = true;
        }
    }
    if (!reused) {
        rc = ensure_e1(m);
        if (rc) return rc;
        // (Replaying the three launches below as an instantiated hipGraph was measured on ROCm 7.2: 6-9 us SLOWER per
        // evaluation than launching them on the stream, at 79 800 pairs and at a 9 975-pair shard.)
        // (Round 3: pipelining a full refit over the two streams -- first eighth of the range k01 -> pairs on the context's
        // stream, the rest k01 -> pairs on the side stream beside it -- was measured too: 0.392 against 0.370 ms per step;
        // two concurrent pair-kernel launches cost more than the hidden 23 us of k01_kernel.)
        HIP_TRY(ecc_launch_k01(&p, ctx->stream));
        if (m->record_reuse && !K01_d && count > 0) {
            m->rec_Ps.assign(Pcur, Pcur + 12 * n);
            m->rec_first = first;
            m->rec_count = count;
            m->rec_n_views = (int)n;
            m->rec_mode = mode;
            m->rec_radius = p.object_radius_mm;
            m->rec_dkappa = p.dkappa_user;
            m->rec_tol = p.economise_tol;
        }
    }
    if (!pairs_launched) {
        if (ctx->timing) HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
        HIP_TRY(ecc_launch_pairs(&p, ctx->stream));
        if (ctx->timing) {
            HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
            ctx->ev_valid[0] = true;
        }
    }
    if (sum_d) {
        if (count > 0) HIP_TRY(ecc_launch_sum_pairs(pair_values_d, count, sum_d, m->sum_scratch_d, ctx->stream));
        else if (sum_d == m->sum_h_dev) std::memset(m->sum_h, 0, sizeof(double));  // empty shard: nothing is launched
        else HIP_TRY(hipMemsetAsync(sum_d, 0, sizeof(double), ctx->stream));
    }
    m->rec_valid = m->record_reuse && !K01_d && count > 0;
    return ECC_OK;
}

// Pair values of [first, first + count) into a device array the metric keeps, their float64 sum to sum_d.
// With ecc_metric_set_incremental: when this range was evaluated before with the same parameters and few matrices have
// changed since, only the pairs that contain a changed view are re-evaluated (index-list launch that writes each value
// into its slot) and the sum kernel runs over the kept array -- every value, and therefore the sum, is bit-identical to
// a full evaluation: a pair's value depends only on its two matrices, its two dtrs and the parameters, the sampling mode
// is the one the full range resolves to, and the sum's order is fixed.
int evaluate_cached(ecc_metric* m, int64_t first, int64_t count, double* sum_d, float** vals_out)
{
    ecc_ctx* ctx = m->ctx;
    const int64_t n = m->n_views;
    if (n < 1) return fail(ECC_ERR_INVALID_ARGUMENT, "projection matrices have not been set");
    if ((int64_t)m->dtrs.size() < n) return fail(ECC_ERR_INVALID_ARGUMENT, "fewer Radon intermediates than projection matrices");
    if (first < 0 || count < 0 || first + count > n * (n - 1) / 2)
        return fail(ECC_ERR_INVALID_ARGUMENT, "pair range outside [0, n(n-1)/2)");
    int rc = ensure_capacity(&m->cache_values_d, &m->cache_capacity, count > 0 ? count : 1, ctx->stream);
    if (rc) return rc;
    *vals_out = m->cache_values_d;
    const double* Pcur = m->Ps_h[m->set_generation & 1];
    double radius = 0;
    ecc_metric_get_object_radius(m, &radius);  // the automatic radius follows the first matrix
    const bool same = m->cache_valid && m->cache_first == first && m->cache_count == count && m->cache_n_views == (int)n &&
                      m->cache_use_corr == m->use_corr && m->cache_sampling == m->sampling && m->cache_radius == radius &&
                      m->cache_dkappa == m->dkappa && (int64_t)m->cache_Ps.size() == 12 * n;
    m->cache_valid = false;  // until everything below is enqueued
    if (same && count > 0) {
        std::vector<int>& changed = m->scratch_changed;
        changed.clear();
        for (int64_t v = 0; v < n; ++v)
            if (std::memcmp(Pcur + 12 * v, m->cache_Ps.data() + 12 * v, sizeof(double) * 12) != 0) changed.push_back((int)v);
        if ((int64_t)changed.size() * 4 <= n) {  // c of n views changed: 1 - (1 - c/n)^2 of the pairs, at most 44 %
            std::vector<char>& is_changed = m->scratch_is_changed;
            is_changed.assign((size_t)n, 0);
            for (int v : changed) is_changed[v] = 1;
            std::vector<int32_t>&idx = m->scratch_idx, &slots = m->scratch_slots;
            idx.clear();
            slots.clear();
            for (int v : changed)
                for (int64_t u = 0; u < n; ++u) {
                    if (u == v || (is_changed[u] && u < v)) continue;  // a pair of two changed views once
                    const int64_t i = u < v ? u : v, j = u < v ? v : u;
                    const int64_t ij = i * n - i * (i + 1) / 2 + (j - i - 1);  // get_ij order
                    if (ij < first || ij >= first + count) continue;
                    idx.insert(idx.end(), {(int32_t)i, (int32_t)j, (int32_t)i, (int32_t)j});
                    slots.push_back((int32_t)(ij - first));
                }
            const int64_t L = (int64_t)slots.size();
            if (L > 0) {
                if (m->list_capacity < L) {
                    if (m->list_h) HIP_TRY(hipHostFree(m->list_h));  // the stream is idle: evaluations are synchronous
                    m->list_h = nullptr;
                    m->list_capacity = 0;
                    const int64_t cap = std::max<int64_t>(2 * L, 1024);
                    HIP_TRY(hipHostMalloc((void**)&m->list_h, sizeof(int32_t) * 5 * cap, hipHostMallocMapped));
                    HIP_TRY(hipHostGetDevicePointer((void**)&m->list_h_dev, m->list_h, 0));
                    m->list_capacity = cap;
                }
                std::memcpy(m->list_h, idx.data(), sizeof(int32_t) * 4 * L);
                std::memcpy(m->list_h + 4 * L, slots.data(), sizeof(int32_t) * L);
                EccPairParams p;
                rc = fill_pair_params(m, &p, n * (n - 1) / 2);  // the sampling mode of the full evaluation
                if (rc) return rc;
                m->rec_valid = false;  // the list's records overwrite the kept ones
                rc = ensure_capacity(&m->records_d, &m->records_capacity, L, ctx->stream);
                if (rc) return rc;
                p.indices = m->list_h_dev;  // read over PCIe inside k01_kernel: 20 bytes per pair, no copy command
                p.value_slots = m->list_h_dev + 4 * L;
                p.first = 0;
                p.count = L;
                p.pair_values = m->cache_values_d;
                p.records = m->records_d;
                HIP_TRY(ecc_launch_k01(&p, ctx->stream));
                if (ctx->timing) HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
                HIP_TRY(ecc_launch_pairs(&p, ctx->stream));
                if (ctx->timing) {
                    HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
                    ctx->ev_valid[0] = true;
                }
            }
            HIP_TRY(ecc_launch_sum_pairs(m->cache_values_d, count, sum_d, m->sum_scratch_d, ctx->stream));
            for (int v : changed) std::memcpy(m->cache_Ps.data() + 12 * v, Pcur + 12 * v, sizeof(double) * 12);
            m->last_evaluated_pairs = L;
            m->cache_valid = true;
            return ECC_OK;
        }
    }
    rc = launch_range(m, first, count, m->cache_values_d, nullptr, nullptr, sum_d);
    if (rc) return rc;
    m->cache_Ps.assign(Pcur, Pcur + 12 * n);
    m->cache_first = first;
    m->cache_count = count;
    m->cache_n_views = (int)n;
    m->cache_use_corr = m->use_corr;
    m->cache_sampling = m->sampling;
    m->cache_radius = radius;
    m->cache_dkappa = m->dkappa;
    m->last_evaluated_pairs = count;
    m->cache_valid = true;
    return ECC_OK;
}

}  // namespace

ECC_EXPORT int ecc_metric_evaluate_range_async(ecc_metric* m, int64_t first, int64_t count, float* pair_values_d,
                                               double* sum_d)
{
    if (!m) return fail(ECC_ERR_INVALID_ARGUMENT, "metric is null");
    int rc = set_device(m->ctx);
    if (rc) return rc;
    float* vals = pair_values_d;
    if (!vals) {
        rc = ensure_capacity(&m->pair_values_d, &m->pair_capacity, count > 0 ? count : 1, m->ctx->stream);
        if (rc) return rc;
        vals = m->pair_values_d;
    }
    return launch_range(m, first, count, vals, nullptr, nullptr, sum_d);
}

ECC_EXPORT int ecc_metric_publish_scalar(ecc_metric* m, const double* value_d)
{
    if (!m || !value_d) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    int rc = set_device(m->ctx);
    if (rc) return rc;
    arm_result(m);
    HIP_TRY(ecc_launch_publish_scalar(value_d, m->sum_h_dev, m->ctx->stream));
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_wait_scalar(ecc_metric* m, double* value)
{
    if (!m || !value) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    int rc = set_device(m->ctx);
    if (rc) return rc;
    HIP_TRY(wait_result(m, m->ctx->stream, value));
    m->done_generation = m->set_generation;  // the publishing kernel is ordered behind everything the metric launched
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_evaluate_range(ecc_metric* m, int64_t first, int64_t count, float* pair_values,
                                         double* partial_sum)
{
    if (!m || !partial_sum) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    ecc_ctx* ctx = m->ctx;
    int rc = set_device(ctx);
    if (rc) return rc;
    rc = ensure_capacity(&m->pair_values_d, &m->pair_capacity, count > 0 ? count : 1, ctx->stream);
    if (rc) return rc;
    arm_result(m);
    float* vals_d = m->pair_values_d;
    if (m->incremental) rc = evaluate_cached(m, first, count, m->sum_h_dev, &vals_d);
    else {
        rc = launch_range(m, first, count, m->pair_values_d, nullptr, nullptr, m->sum_h_dev, /*synchronous=*/true);
        m->last_evaluated_pairs = count;
    }
    if (rc) return rc;
    if (pair_values && count > 0) {
        HIP_TRY(hipMemcpyAsync(pair_values, vals_d, sizeof(float) * count, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(wait_stream_spin(ctx->stream));  // the copy has to land too
    }
    HIP_TRY(wait_sum(m, partial_sum));
    // an empty shard launches no kernel behind e1_kernel: its result slot says nothing about the stream
    if (count == 0) HIP_TRY(wait_stream_spin(ctx->stream));
    m->done_generation = m->set_generation;
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_evaluate_all(ecc_metric* m, float* cost_nxn, double* mean)
{
    if (!m || !mean) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    ecc_ctx* ctx = m->ctx;
    int rc = set_device(ctx);
    if (rc) return rc;
    const int64_t n = m->n_views;
    const int64_t n_pairs = n * (n - 1) / 2;
    if (n < 2) return fail(ECC_ERR_INVALID_ARGUMENT, "need at least two views (the reference divides 0/0 here)");
    rc = ensure_capacity(&m->pair_values_d, &m->pair_capacity, n_pairs, ctx->stream);
    if (rc) return rc;
    float* cost_d = nullptr;
    if (cost_nxn) {
        if (m->cost_capacity < n * n) {
            if (m->cost_d) {
                HIP_TRY(hipStreamSynchronize(ctx->stream));
                HIP_TRY(hipFree(m->cost_d));
                m->cost_d = nullptr;
            }
            HIP_TRY(hipMalloc((void**)&m->cost_d, sizeof(float) * n * n));
            m->cost_capacity = (int)(n * n);
        }
        cost_d = m->cost_d;
        // upload the caller's image so that untouched entries survive, ref: ...RadonIntermediate.cpp:183
        HIP_TRY(hipMemcpyAsync(cost_d, cost_nxn, sizeof(float) * n * n, hipMemcpyHostToDevice, ctx->stream));
    }
    arm_result(m);
    if (m->incremental && !cost_nxn) {  // with a cost image every pair is written anyway
        float* vals_d = nullptr;
        rc = evaluate_cached(m, 0, n_pairs, m->sum_h_dev, &vals_d);
    } else {
        rc = launch_range(m, 0, n_pairs, m->pair_values_d, cost_d, nullptr, m->sum_h_dev, /*synchronous=*/true);  // the sum lands in pinned host memory
        m->last_evaluated_pairs = n_pairs;
    }
    if (rc) return rc;
    if (cost_nxn) {
        HIP_TRY(hipMemcpyAsync(cost_nxn, cost_d, sizeof(float) * n * n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(wait_stream_spin(ctx->stream));
    }
    double sum = 0.0;
    HIP_TRY(wait_sum(m, &sum));
    m->done_generation = m->set_generation;
    *mean = sum / (double)n_pairs;  // ref: ...RadonIntermediate.cpp:224 (all weights are 1)
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_evaluate_pairs(ecc_metric* m, const int32_t* idx4, int n_pairs, float* out, double* mean)
{
    if (!m || !idx4 || !mean) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (n_pairs < 1) return fail(ECC_ERR_INVALID_ARGUMENT, "empty index list (the reference divides 0/0 here)");
    ecc_ctx* ctx = m->ctx;
    int rc = set_device(ctx);
    if (rc) return rc;
    // range check in every build; the reference only does it under _DEBUG (...RadonIntermediate.cpp:248-275)
    const int nP = m->n_views, nD = (int)m->dtrs.size();
    for (int q = 0; q < n_pairs; ++q) {
        const int32_t* t = idx4 + 4 * (size_t)q;
        if (t[0] < 0 || t[0] >= nP || t[1] < 0 || t[1] >= nP || t[2] < 0 || t[2] >= nD || t[3] < 0 || t[3] >= nD)
            return fail(ECC_ERR_INVALID_ARGUMENT, "index array contains invalid indices");
    }
    rc = ensure_capacity(&m->indices_d, &m->indices_capacity, (int64_t)4 * n_pairs, ctx->stream);
    if (rc) return rc;
    rc = ensure_capacity(&m->pair_values_d, &m->pair_capacity, n_pairs, ctx->stream);
    if (rc) return rc;
    EccPairParams p;
    rc = fill_pair_params(m, &p, n_pairs, /*need_e1=*/false);
    if (rc) return rc;
    {   // few pairs: ONE launch; the list and the values travel through pinned memory, no copy commands
        EccPairParams q = p;
        q.first = 0;
        q.count = n_pairs;
        q.pair_values = m->pair_values_d;
        bool taken = false;
        arm_result(m);
        rc = try_small_eval(m, q, idx4, &taken);
        if (rc) return rc;
        if (taken) {
            double sum = 0.0;
            HIP_TRY(wait_sum(m, &sum));
            m->done_generation = m->set_generation;
            if (out) std::memcpy(out, m->svals_h, sizeof(float) * (size_t)n_pairs);
            *mean = sum / (double)n_pairs;
            return ECC_OK;
        }
    }
    rc = ensure_e1(m);
    if (rc) return rc;
    m->rec_valid = false;  // the list's records overwrite the kept ones
    rc = ensure_capacity(&m->records_d, &m->records_capacity, n_pairs, ctx->stream);
    if (rc) return rc;
    // Up to 32 768 pairs the list is read by k01_kernel straight from pinned host memory (16 bytes per pair over PCIe) and the
    // values come back through the sum kernel, which stores what it loads into pinned memory in front of the result:
    // no copy commands (they cost an index-list evaluation 25 us: 512 pairs 57 -> ~30 us).
    const bool pinned = n_pairs < 32768;
    if (pinned) {
        rc = ensure_small_host_buffers(m, n_pairs, n_pairs);
        if (rc) return rc;
        std::memcpy(m->sidx_h, idx4, sizeof(int32_t) * 4 * (size_t)n_pairs);
        std::atomic_thread_fence(std::memory_order_seq_cst);
        p.indices = m->sidx_h_dev;
    } else {
        HIP_TRY(hipMemcpyAsync(m->indices_d, idx4, sizeof(int32_t) * 4 * n_pairs, hipMemcpyHostToDevice, ctx->stream));
        p.indices = m->indices_d;
    }
    p.first = 0;
    p.count = n_pairs;
    p.pair_values = m->pair_values_d;
    p.records = m->records_d;
    HIP_TRY(ecc_launch_k01(&p, ctx->stream));
    if (ctx->timing) HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
    HIP_TRY(ecc_launch_pairs(&p, ctx->stream));
    if (ctx->timing) {
        HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
        ctx->ev_valid[0] = true;
    }
    arm_result(m);
    if (pinned) {
        HIP_TRY(ecc_launch_sum_pairs_to_host(m->pair_values_d, n_pairs, m->sum_h_dev, out ? m->svals_h_dev : nullptr, ctx->stream));
    } else {
        HIP_TRY(ecc_launch_sum_pairs(m->pair_values_d, n_pairs, m->sum_h_dev, m->sum_scratch_d, ctx->stream));
        if (out) {
            HIP_TRY(hipMemcpyAsync(out, m->pair_values_d, sizeof(float) * n_pairs, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(wait_stream_spin(ctx->stream));
        }
    }
    double sum = 0.0;
    HIP_TRY(wait_result(m, ctx->stream, &sum));
    if (pinned && out) {
        std::atomic_thread_fence(std::memory_order_acquire);
        std::memcpy(out, m->svals_h, sizeof(float) * (size_t)n_pairs);
    }
    m->done_generation = m->set_generation;
    *mean = sum / (double)n_pairs;
    return ECC_OK;
}

// The reference's launcher seam for the metric (ref: EpipolarConsistencyRadonIntermediate.cpp:16-37 epipolarConsistency(...),
// .cu:300-409): everything in device memory owned by the caller, the per-view geometry already made by the caller's host
// class (culaut, ref: ...RadonIntermediate.cpp:134-163).  indices_d == null: all n (n - 1) / 2 pairs, out_d is the n x n
// cost image (entry i + j n, i < j, overwritten; the rest untouched); else out_d receives num_pairs values.  K01s_d
// (nullable): the 16 floats per pair the reference keeps between its two kernels.  Returns after the stream has run (the
// reference synchronises the device after each of its kernels).  No E1, no kept records, no host result: the caller's
// epilogue reads out_d back and forms the mean (ref: ...RadonIntermediate.cpp:197-224).
ECC_EXPORT int ecc_metric_evaluate_external(ecc_metric* m, int num_Ps, const float* Cs_d, const float* PinvTs_d, int num_pairs,
                                            const int32_t* indices_d, float* K01s_d, float* out_d, float object_radius_mm,
                                            float dkappa, int use_corr)
{
    if (!m || !Cs_d || !PinvTs_d || !out_d) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (num_Ps < 2 || num_pairs < 1) return fail(ECC_ERR_INVALID_ARGUMENT, "need at least two views and one pair");
    if ((int)m->dtrs.size() < num_Ps && !indices_d)
        return fail(ECC_ERR_INVALID_ARGUMENT, "fewer Radon intermediates than projection matrices");
    if (!indices_d && (int64_t)num_pairs != (int64_t)num_Ps * (num_Ps - 1) / 2)
        return fail(ECC_ERR_INVALID_ARGUMENT, "all-pairs form: num_pairs must be n (n - 1) / 2");
    ecc_ctx* ctx = m->ctx;
    int rc = set_device(ctx);
    if (rc) return rc;
    // parameters of this call only (the metric's own are restored below)
    const double keep_radius = m->object_radius_mm, keep_dkappa = m->dkappa;
    const int keep_corr = m->use_corr, keep_n = m->n_views;
    m->object_radius_mm = object_radius_mm > 0 ? object_radius_mm : 1e-30;  // the seam has no "automatic": the caller passes its radius
    m->dkappa = dkappa;
    m->use_corr = use_corr;
    m->n_views = num_Ps;
    EccPairParams p;
    rc = fill_pair_params(m, &p, num_pairs, /*need_e1=*/false);
    m->object_radius_mm = keep_radius;
    m->dkappa = keep_dkappa;
    m->use_corr = keep_corr;
    m->n_views = keep_n;
    if (rc) return rc;
    p.object_radius_mm = object_radius_mm;
    p.Cs = Cs_d;
    p.PinvTs = PinvTs_d;
    p.n_views = num_Ps;
    m->rec_valid = false;  // the records below overwrite the kept ones
    rc = ensure_capacity(&m->records_d, &m->records_capacity, num_pairs, ctx->stream);
    if (rc) return rc;
    p.indices = indices_d;
    p.first = 0;
    p.count = num_pairs;
    p.records = m->records_d;
    p.K01_out = K01s_d;
    if (indices_d) p.pair_values = out_d;
    else p.cost = out_d;
    HIP_TRY(ecc_launch_k01(&p, ctx->stream));
    HIP_TRY(ecc_launch_pairs(&p, ctx->stream));
    HIP_TRY(wait_stream_spin(ctx->stream));
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_debug_K01(ecc_metric* m, int64_t first, int64_t count, float* K01s)
{
    if (!m || !K01s) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (count < 1) return fail(ECC_ERR_INVALID_ARGUMENT, "empty range");
    ecc_ctx* ctx = m->ctx;
    int rc = set_device(ctx);
    if (rc) return rc;
    rc = ensure_capacity(&m->pair_values_d, &m->pair_capacity, count, ctx->stream);
    if (rc) return rc;
    rc = ensure_capacity(&m->K01_d, &m->K01_capacity, 16 * count, ctx->stream);
    if (rc) return rc;
    rc = launch_range(m, first, count, m->pair_values_d, nullptr, m->K01_d, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(K01s, m->K01_d, sizeof(float) * 16 * count, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
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

// ---- debug: the fitted sample-coordinate polynomials ---------------------------------------------
ECC_EXPORT int ecc_metric_debug_polynomials(ecc_metric* m, int64_t first, int64_t count, float* out)
{
    static_assert(ECC_POLY_RECORD_FLOATS == 4 + 2 * (ECC_POLY_DEG + 3) + 2 * (ECC_POLY_DEG + 2), "header and layout disagree");
    if (!m || !out) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (count < 1) return fail(ECC_ERR_INVALID_ARGUMENT, "empty range");
    ecc_ctx* ctx = m->ctx;
    int rc = set_device(ctx);
    if (rc) return rc;
    rc = ensure_capacity(&m->pair_values_d, &m->pair_capacity, count, ctx->stream);
    if (rc) return rc;
    rc = launch_range(m, first, count, m->pair_values_d, nullptr, nullptr, nullptr);  // fills m->records_d[0..count)
    if (rc) return rc;
    std::vector<EccPairRecord> recs((size_t)count);
    HIP_TRY(hipMemcpyAsync(recs.data(), m->records_d, sizeof(EccPairRecord) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int64_t q = 0; q < count; ++q) {
        const EccPairRecord& r = recs[(size_t)q];
        float* o = out + (size_t)q * ECC_POLY_RECORD_FLOATS;
        *o++ = (float)r.poly_ok;
        *o++ = r.x_scale;
        *o++ = r.fold[0] ? 1.f : 0.f;
        *o++ = r.fold[1] ? 1.f : 0.f;
        for (int v = 0; v < 2; ++v)
            for (int k = 0; k < ECC_POLY_DEG + 3; ++k) *o++ = r.ca[v][k];
        for (int v = 0; v < 2; ++v)
            for (int k = 0; k < ECC_POLY_DEG + 2; ++k) *o++ = r.cd[v][k];
    }
    return ECC_OK;
}

/* Experiments (ECC_SMALL_DEBUG=1): the wall-clock stamps (100 MHz) of the last small_eval_kernel launch, 4 per workgroup. */
ECC_EXPORT int ecc_debug_small_stamps(unsigned long long* out, int n_blocks)
{
    if (!g_small_dbg || !out || n_blocks < 1 || n_blocks > 4096) return ECC_ERR_INVALID_ARGUMENT;
    if (hipDeviceSynchronize() != hipSuccess) return ECC_ERR_HIP;
    return hipMemcpy(out, g_small_dbg, sizeof(unsigned long long) * 4 * n_blocks, hipMemcpyDeviceToHost) == hipSuccess ? ECC_OK : ECC_ERR_HIP;
}
