// ecc_direct_api.hip -- MetricDirect / FBCC behind the C ABI (host code only; the kernels are direct_kernel.hip).
// ref: EpipolarConsistencyDirect.{h,cpp,cu}, RectifiedFBCC.h.
#include "ecc_capi_internal.h"

#define ECC_EXPORT extern "C" __attribute__((visibility("default")))

using namespace ecc_internal;

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
