// ecc_metric_api.hip -- MetricRadonIntermediate's objects behind the C ABI (host code only): creation, Radon intermediates and
// their row-paired copies, projection matrices and parameters (ref: EpipolarConsistencyRadonIntermediate.cpp:53-163,
// EpipolarConsistency.cpp:35-104), evaluateForImagePair (E7, .cpp:324-393), the Metric base class's free helpers and the
// cost-balanced shards of the pair range.  The evaluation paths themselves are in ecc_evaluate.hip.
#include "ecc_capi_internal.h"

#define ECC_EXPORT extern "C" __attribute__((visibility("default")))

using namespace ecc_internal;

// ---- metric ------------------------------------------------------------------------------------
namespace {
// The row-quad copies of all the metric's Radon intermediates: allocation, table, build launch on the context's stream (ordered
// in front of whatever samples them).  On failure nothing is left behind.
hipError_t build_quad_copies(ecc_metric* m)
{
    const int n_dtrs = (int)m->dtrs.size();
    std::vector<const float*>& qtable = m->quads_table_h;  // (lives with the metric: the asynchronous upload below may read it after this returns)
    qtable.assign((size_t)n_dtrs, nullptr);
    hipError_t e = hipMalloc((void**)&m->quads_table_d, sizeof(float*) * n_dtrs);
    if (e == hipSuccess) e = hipMalloc((void**)&m->quads_d, sizeof(float) * (size_t)m->quad_floats * n_dtrs);
    for (int k = 0; k < n_dtrs && e == hipSuccess; ++k) qtable[k] = m->quads_d + (size_t)m->quad_floats * k;
    if (e == hipSuccess) e = hipMemcpyAsync(m->quads_table_d, qtable.data(), sizeof(float*) * n_dtrs, hipMemcpyHostToDevice, m->ctx->stream);
    if (e == hipSuccess) e = ecc_launch_build_quad(m->dtr_table_d, m->quads_d, m->quad_floats, n_dtrs, m->n_alpha + 1, m->pitch, m->ctx->stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (m->quads_d) (void)hipFree(m->quads_d);
        if (m->quads_table_d) (void)hipFree((void*)m->quads_table_d);
        m->quads_d = nullptr;
        m->quads_table_d = nullptr;
    }
    return e;
}
}  // namespace

namespace ecc_internal {
// ECC_QUAD_COPIES_AUTO, decided once per metric by its first evaluation of at least 32 768 pairs (launch_range): the copies cost
// 4x the stack and only the pairs with kappa_max > pi/4 read them (in practice those whose baseline passes through the object,
// kappa_max = pi/2: 3.5 % of a 200-degree short scan's pairs, none of a 90-degree scan's).  Built when at least 2 % of the
// current matrices' pairs are such pairs AND all copies together fit a quarter of the free device memory; an allocation that
// fails after all means "none" (the same bits either way).  kappa_max from the source positions alone (ref: computeK01,
// EpipolarConsistencyCommon.hxx:115-123,137-145), float64 on the host: ~0.3 ms for 79 800 pairs, once.
void decide_quad_copies(ecc_metric* m)
{
    if (m->quads_decided) return;
    m->quads_decided = true;
    const int64_t n = m->n_views;
    if (n < 2 || (int64_t)m->dtrs.size() < n) return;
    double radius = 0;
    ecc_metric_get_object_radius(m, &radius);
    const double* Ps = m->Ps_h[m->set_generation & 1];
    std::vector<double> C(4 * (size_t)n);
    for (int64_t v = 0; v < n; ++v) {
        float c4[4];
        ecc_host::source_position(Ps + 12 * (size_t)v, c4);
        for (int k = 0; k < 4; ++k) C[4 * (size_t)v + k] = c4[k];
    }
    int64_t wide = 0, looked = 0;
    const double sin_quarter = 0.70710678118654752;  // kappa_max > pi/4  <=>  radius / dist > sin(pi/4)
    // (a sample of the pairs when there are millions: every `step`-th partner of every view, at most ~500 000 pairs)
    const int64_t step = std::max<int64_t>(1, n * (n - 1) / 2 / 500000);
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = i + 1 + (i % step); j < n; j += step) {
            ++looked;
            const double *a = &C[4 * (size_t)i], *b = &C[4 * (size_t)j];
            const double B01 = a[0] * b[1] - a[1] * b[0], B02 = a[0] * b[2] - a[2] * b[0], B03 = a[0] * b[3] - a[3] * b[0];
            const double B12 = a[1] * b[2] - a[2] * b[1], B13 = a[1] * b[3] - a[3] * b[1], B23 = a[2] * b[3] - a[3] * b[2];
            const double s2 = std::sqrt(B12 * B12 + B02 * B02 + B01 * B01), s3 = std::sqrt(B03 * B03 + B13 * B13 + B23 * B23);
            if (!(s2 * sin_quarter > radius * s3)) ++wide;  // dist = s2 / s3 (also NaN geometry)
        }
    if (wide * 50 < looked) return;  // under 2 %
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    if ((uint64_t)m->quad_floats * 4u * (uint64_t)m->dtrs.size() > (uint64_t)free_b / 4u) return;
    (void)build_quad_copies(m);  // (failure: no copies; the error state is cleared)
}
}  // namespace ecc_internal

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
    // row-quad copies (ecc_ctx_set_quad_copies; 4x the slab memory, see pair_accumulate): offsets must fit 32 bits.
    // ECC_QUAD_COPIES_ON: built here.  ECC_QUAD_COPIES_AUTO (default): decided by the first large evaluation, when the matrices
    // say whether the scan has pairs that read them at all (ecc_internal::decide_quad_copies); _OFF: never.
    m->quad_floats = (int64_t)((m->n_alpha + 1 + 3) / 4) * m->pitch * 16;
    m->quads_possible = m->quad_floats * 4 < ((int64_t)1 << 32);
    m->quads_decided = !(m->quads_possible && ctx->quad_copies == ECC_QUAD_COPIES_AUTO);
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
    if (e == hipSuccess && m->quads_possible && ctx->quad_copies == ECC_QUAD_COPIES_ON) e = build_quad_copies(m);  // (behind the table's upload)
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
    if (m) ecc_mark_busy(m);
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
    ecc_mark_busy(m);
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
    if (m->pose_h) (void)hipHostFree(m->pose_h);
    if (m->pose_PinvTs_d) (void)hipFree(m->pose_PinvTs_d);
    if (m->pose_Cs_d) (void)hipFree(m->pose_Cs_d);
    if (m->pose_idx_d) (void)hipFree(m->pose_idx_d);
    if (m->pose_records_d) (void)hipFree(m->pose_records_d);
    if (m->pose_values_d) (void)hipFree(m->pose_values_d);
    if (m->pose_partial_d) (void)hipFree(m->pose_partial_d);
    if (m->pose_lists_d) (void)hipFree(m->pose_lists_d);
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

// Device memory the metric owns right now (include/ecc_hip.h): the two sampling copies of the Radon-intermediate stack and
// everything else (geometry, records, pair values, cost image, lists, the pose batch's scratch).
ECC_EXPORT int ecc_metric_device_bytes(const ecc_metric* m, int64_t* paired_bytes, int64_t* quad_bytes, int64_t* other_bytes)
{
    if (!m) return fail(ECC_ERR_INVALID_ARGUMENT, "metric is null");
    const int64_t n = (int64_t)m->dtrs.size();
    const int64_t paired = m->paired_d ? (int64_t)(m->n_alpha + 1) * m->pitch * 2 * 4 * n : 0;
    const int64_t quads = m->quads_d ? m->quad_floats * 4 * n : 0;
    int64_t other = 0;
    other += (int64_t)sizeof(float*) * n * (2 + (m->quads_table_d ? 1 : 0));
    other += (int64_t)m->geom_capacity * (16 * (int64_t)sizeof(float) + 12 * (int64_t)sizeof(double));
    other += m->pair_capacity * 4 + (int64_t)m->cost_capacity * 4 + m->indices_capacity * 4 + m->K01_capacity * 4;
    other += m->records_capacity * (int64_t)sizeof(EccPairRecord) + m->cache_capacity * 4;
    other += m->pose_PinvTs_capacity * 4 + m->pose_Cs_capacity * 4 + m->pose_idx_capacity * 4 + m->pose_values_capacity * 4;
    other += m->pose_records_capacity * (int64_t)sizeof(EccPairRecord) + m->pose_partial_capacity * 8 + m->pose_lists_capacity * 4;
    other += (int64_t)ecc_sum_scratch_bytes() + 8;
    if (paired_bytes) *paired_bytes = paired;
    if (quad_bytes) *quad_bytes = quads;
    if (other_bytes) *other_bytes = other;
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
    if (m) ecc_mark_busy(m);
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
