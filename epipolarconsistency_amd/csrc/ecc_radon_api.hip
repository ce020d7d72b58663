// ecc_radon_api.hip -- the Radon-intermediate part of the C ABI (host code only): ecc_radon_* and ecc_dtr_* of include/ecc_hip.h.
// ref: RadonIntermediate ctor / compute / readback / replace (LibEpipolarConsistency/RadonIntermediate.cpp:17-31,105-123,148-163,
// 198-211) and its launcher computeDerivLineIntegrals (RadonIntermediate.cpp:12, .cu:149-170).
#include "ecc_capi_internal.h"

#define ECC_EXPORT extern "C" __attribute__((visibility("default")))

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
