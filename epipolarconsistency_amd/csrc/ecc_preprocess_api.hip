// ecc_preprocess_api.hip -- projection pre-processing behind the C ABI (host code only; the kernel is preprocess_kernel.hip).
// ref: Gui/PreProccess.cpp:57-166 (scale / bias / -log, border zero + feather, flips, low-pass, cosine weighting).
#include "ecc_capi_internal.h"

#define ECC_EXPORT extern "C" __attribute__((visibility("default")))

using namespace ecc_internal;

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
