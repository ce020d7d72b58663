// ecc_evaluate.hip -- the evaluation paths of MetricRadonIntermediate behind the C ABI (host code only; the object layouts
// and shared helpers are in ecc_capi_internal.h, the metric's objects in ecc_metric_api.hip).
//
// ref: MetricRadonIntermediate::evaluate(float*), evaluate(indices, out) (EpipolarConsistencyRadonIntermediate.cpp:166-225,
// 267-322) and their launcher epipolarConsistency(...) (.cpp:16-37, .cu:300-409).  Here: the parameters of a launch
// (fill_pair_params), the stream-ordered launches over a pair range with the kept records and the two-stream refit
// (launch_range), the one launch for small evaluations (try_small_eval), the pose-delta path (evaluate_cached), and the
// exported evaluate calls on top of them.
#include "ecc_capi_internal.h"

#define ECC_EXPORT extern "C" __attribute__((visibility("default")))

using namespace ecc_internal;

namespace {

double pose_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

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
}  // namespace
namespace ecc_internal {
int fill_pair_params(ecc_metric* m, EccPairParams* p, int64_t mode_count, bool need_e1)
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
    p->economise_tol = m->economise_tol;  // ECC_POLY_ECONOMISE_TOL_BINS unless ecc_debug_set_poly_tolerance changed it
    return ECC_OK;
}
}  // namespace ecc_internal
namespace {

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

// The pair values a kernel hands to the host through pinned memory (small_eval_kernel: one system-scope store per
// workgroup; sum_pairs_kernel with values_host) are ordered in front of the word the host polls by the device's own
// drains and barriers, not by anything the HIP memory model promises across workgroups.  So the hand-over checks itself:
// the host arms every slot it expects with a NaN payload no kernel stores, and after the polled word has arrived it
// accepts the values only once no slot is armed any more (bounded wait, then an error -- never a stale value).
constexpr uint32_t ECC_VALUE_ARMED = 0x7fc0ecc1u;
void arm_values(ecc_metric* m, int64_t count)
{
    volatile uint32_t* v = reinterpret_cast<volatile uint32_t*>(m->svals_h);
    for (int64_t k = 0; k < count; ++k) v[k] = ECC_VALUE_ARMED;
}
hipError_t wait_values(ecc_metric* m, int64_t count)
{
    const volatile uint32_t* v = reinterpret_cast<const volatile uint32_t*>(m->svals_h);
    double t0 = 0.0;
    int64_t k = 0;
    for (unsigned spins = 0;; ++spins) {
        while (k < count && v[k] != ECC_VALUE_ARMED) ++k;
        if (k == count) break;
        if ((spins & 0xff) == 0xff) {
            const double t = pose_now();
            if (t0 == 0.0) t0 = t;
            else if (t - t0 > 2.0) {
                const hipError_t e = hipStreamSynchronize(m->ctx->stream);
                if (e != hipSuccess) return e;
                for (k = 0; k < count; ++k)
                    if (v[k] == ECC_VALUE_ARMED) return hipErrorUnknown;  // the kernel has ended and a value never arrived
                break;
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return hipSuccess;
}

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
    const hipError_t ev = wait_values(m, count);
    if (ev != hipSuccess) return ev;
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
// E1 for a launch that takes it in its kernel arguments (small_eval_kernel, k01_patched_kernel): the views whose geometry on
// the device is behind the current matrices go into x->patch_*, computed here with the code e1_kernel compiles
// (ecc_host_geometry.h, bit-identical; ref: ...RadonIntermediate.cpp:134-163); more than ECC_SMALL_PATCH_MAX of them (the
// first call, a new trajectory): e1_kernel, ordered before the launch, and x->patch_count stays 0.
// The host's view of the device arrays (dev_Ps, e1_pending) is NOT updated here: the caller calls commit_patches once the
// launch that stores the patches has been enqueued -- a failure in between must leave the views marked stale.
int small_eval_patches(ecc_metric* m, EccSmallEval* x, bool* from_host)
{
    *from_host = false;
    const int n = m->n_views;
    const double* Pcur = m->Ps_h[m->set_generation & 1];
    std::vector<int>& stale = m->scratch_stale;
    stale.clear();
    const bool dev_known = m->dev_valid && (int64_t)m->dev_Ps.size() == 12 * (int64_t)n;
    if (dev_known && m->e1_pending)
        for (int v = 0; v < n && (int)stale.size() <= ECC_SMALL_PATCH_MAX; ++v)
            if (std::memcmp(Pcur + 12 * v, m->dev_Ps.data() + 12 * v, sizeof(double) * 12) != 0) stale.push_back(v);
    if (!dev_known || (int)stale.size() > ECC_SMALL_PATCH_MAX) return ensure_e1(m);
    for (size_t e = 0; e < stale.size(); ++e) {
        const int v = stale[e];
        ecc_host::pinv_transpose(Pcur + 12 * v, x->patch_geo[e]);
        ecc_host::source_position(Pcur + 12 * v, x->patch_geo[e] + 12);
        x->patch_views[e] = v;
    }
    x->patch_count = (int)stale.size();
    *from_host = true;
    return ECC_OK;
}

// After the launch that carries x's patches is in the stream: workgroup 0 of it stores the entries into PinvTs_d / Cs_d.
void commit_patches(ecc_metric* m, const EccSmallEval& x, bool from_host)
{
    if (!from_host) return;  // (E1 went through ensure_e1, which keeps its own books)
    const double* Pcur = m->Ps_h[m->set_generation & 1];
    for (int e = 0; e < x.patch_count; ++e) {
        const int v = x.patch_views[e];
        std::memcpy(m->dev_Ps.data() + 12 * v, Pcur + 12 * v, sizeof(double) * 12);
    }
    m->e1_pending = false;
}

// k01 over p on the context's stream, E1 included: small launches (8 lanes per fit) of a metric with the one-launch path on
// take E1 of up to ECC_SMALL_PATCH_MAX changed views in their kernel arguments (no e1_kernel launch in front, and
// ecc_metric_set_projections does not launch it either); everything else is e1_kernel (if due) + k01_kernel.
int launch_k01_with_e1(ecc_metric* m, const EccPairParams& p)
{
    if (m->small_eval && p.count > 0 && p.count <= ECC_K01_WIDE_MAX_PAIRS && !p.patch_count) {
        EccSmallEval x;
        std::memset(&x, 0, sizeof(x));
        bool from_host = false;
        const int rc = small_eval_patches(m, &x, &from_host);
        if (rc) return rc;
        m->eager_e1 = false;  // the views that change next are patched by the next launch
        EccPairParams q = p;
        q.PinvTs = m->PinvTs_d;
        q.Cs = m->Cs_d;
        if (x.patch_count > 0) HIP_TRY(ecc_launch_k01_patched(&q, &x, m->ctx->stream));
        else HIP_TRY(ecc_launch_k01(&q, m->ctx->stream));
        commit_patches(m, x, from_host);
        return ECC_OK;
    }
    const int rc = ensure_e1(m);
    if (rc) return rc;
    HIP_TRY(ecc_launch_k01(&p, m->ctx->stream));
    return ECC_OK;
}

int try_small_eval(ecc_metric* m, EccPairParams p, const int32_t* idx4_host, bool* taken)
{
    *taken = false;
    int wpp = 0;
    size_t lds = 0;
    if (!m->small_eval || !ecc_small_eval_plan(&p, m->small_max_pairs, &wpp, &lds)) return ECC_OK;

    ecc_ctx* ctx = m->ctx;
    if (!m->small_ticket_d) {
        HIP_TRY(hipMalloc((void**)&m->small_ticket_d, sizeof(unsigned)));
        HIP_TRY(hipMemsetAsync(m->small_ticket_d, 0, sizeof(unsigned), ctx->stream));
    }
    EccSmallEval x;
    std::memset(&x, 0, sizeof(x));
    bool patches_from_host = false;
    {
        const int rcp = small_eval_patches(m, &x, &patches_from_host);
        if (rcp) return rcp;
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
#ifdef ECC_SMALL_STAMPS  // experiment builds only (scripts/exp_small_phases.py): stamps read back by ecc_debug_small_stamps
    static unsigned long long* dbg_d = [] {
        unsigned long long* d = nullptr;
        (void)hipMalloc((void**)&d, sizeof(unsigned long long) * 4 * 4096);
        return d;
    }();
    x.dbg = dbg_d;
    g_small_dbg = dbg_d;
#endif

    arm_values(m, p.count);
    std::atomic_thread_fence(std::memory_order_seq_cst);  // the host's writes to pinned memory before the doorbell
    if (ctx->timing) HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
    HIP_TRY(ecc_launch_small_eval(&p, &x, ctx->stream));
    commit_patches(m, x, patches_from_host);
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
}  // namespace
namespace ecc_internal {
int launch_range(ecc_metric* m, int64_t first, int64_t count, float* pair_values_d, float* cost_d, float* K01_d,
                 double* sum_d, bool synchronous)
{
    ecc_ctx* ctx = m->ctx;
    const bool was_quiet = m->quiet;  // nothing of this metric's is pending on either stream (ecc_capi_internal.h)
    ecc_mark_busy(m);
    const int64_t n = m->n_views;
    const int64_t n_pairs = n * (n - 1) / 2;
    if ((int)m->dtrs.size() < m->n_views)
        return fail(ECC_ERR_INVALID_ARGUMENT, "fewer Radon intermediates than projection matrices");
    if (first < 0 || count < 0 || first + count > n_pairs)
        return fail(ECC_ERR_INVALID_ARGUMENT, "pair range outside [0, n(n-1)/2)");
    // ECC_QUAD_COPIES_AUTO: the first evaluation of a large pair set decides, from the matrices, whether the row-quad copies are
    // worth their memory (stream-ordered in front of the launches below; the same bits either way)
    if (!m->quads_decided && n_pairs >= 32768 && count > 0) decide_quad_copies(m);  // (the EVALUATION's size: every shard of it decides alike)
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
    ecc_stamp(m, 3);
    m->stamps[4] = m->stamps[5] = 0.0;
    if (rec_match) {
        // (Round 5, measured and dropped: the changed views found in ONE pass by ecc_metric_set_projections while it stages the
        // matrices, with generation counters saying when that list is what the kept records and the device geometry differ
        // by, and the fork event skipped when hipStreamQuery finds the stream idle -- A/B/A/B on one box the step was 2-4 us
        // SLOWER: the three passes over 38 KB below cost less than a microsecond, and the stream is usually NOT yet reported
        // idle when the next step begins -- the polled result arrives before the runtime sees the completion signal --
        // which makes the query the expensive call.  profiles/r05_ab_host_shortcuts.txt)
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
                // A stream of its OWN priority class: the runtime deals the streams of one priority to a few hardware queues in turn,
                // and a process that has created other streams before this one -- an RCCL communicator brings several -- can end up
                // with this stream on the context stream's queue: the list launches then run AFTER the all-pairs launch instead of
                // beside it (step 328 -> 345 us, scripts/experiments/comm_probe.py).  Highest priority: the few hundred pairs of the
                // moved view also get their slots ahead of the big launch's waiting workgroups (lowest: shard step 69 -> 76 us).
                int prio_lo = 0, prio_hi = 0;
                (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
                if (hipStreamCreateWithPriority(&m->side_stream, hipStreamNonBlocking, prio_hi) != hipSuccess ||
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
            // The all-pairs launch (skipping the changed views' pairs) goes out FIRST when it is long (the refit then trickles
            // into the holes its retiring workgroups leave and is through long before it), and AFTER the refit's launch when it
            // is a shard-sized one: beside a 46-us launch of 2 718 workgroups (1.5 rounds of resident ones) the ten workgroups of
            // k01_kernel<16> waited 29 us for room (5 alone), the moved pairs' launch ended after the big one and the sum paid the
            // late cross-stream join (12 us instead of 6) -- kernel trace of scripts/step_fixed_cost.py 8.
            // (One rank's share of an 8-rank job, 10 873 pairs: 70.1 / 71.0 -> 65.9 / 65.9 us per step A/B/A/B; of a 4-rank job,
            // 22 323 pairs, 3.1 rounds: the old order is as good or better -- 101 / 99 against 102 / 114 us.)
            const bool refit_first = split && count < ECC_REFIT_FIRST_MAX_PAIRS;
            auto launch_all_pairs = [&]() -> int {
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
                ecc_stamp(m, 4);
                return ECC_OK;
            };
            if (split) {
                // whatever this metric queued on the context's stream before this call comes first for the side stream too
                // (the side stream reads nothing anybody else writes); nothing to wait for after a synchronous evaluation
                if (!was_quiet) HIP_TRY(hipEventRecord(m->fork_ev, ctx->stream));
                if (!refit_first) {
                    rc = launch_all_pairs();
                    if (rc) return rc;
                }
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
                if (split && !was_quiet) SIDE_TRY(hipStreamWaitEvent(m->side_stream, m->fork_ev, 0));
                SIDE_TRY(ecc_launch_k01(&q, ks));
                if (refit_first) {
                    const int rl = launch_all_pairs();
                    if (rl) {
                        (void)hipStreamSynchronize(m->side_stream);
                        return rl;
                    }
                }
                if (split) {  // the changed pairs' own launch: records and values in their slots
                    q.pair_values = pair_values_d;
                    q.value_slots = q.record_slots;
                    q.beside_another_launch = count >= ECC_BESIDE_ONE_WAVE_MIN_PAIRS ? 2 : 1;
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
            if (refit_first && !pairs_launched) {  // no pair of this range contains a changed view
                rc = launch_all_pairs();
                if (rc) return rc;
            }
            reused = true;
        }
    }
    if (!reused) {
        // (Replaying the three launches below as an instantiated hipGraph was measured on ROCm 7.2: 6-9 us SLOWER per
        // evaluation than launching them on the stream, at 79 800 pairs and at a 9 975-pair shard.)
        // (Round 3: pipelining a full refit over the two streams -- first eighth of the range k01 -> pairs on the context's
        // stream, the rest k01 -> pairs on the side stream beside it -- was measured too: 0.392 against 0.370 ms per step;
        // two concurrent pair-kernel launches cost more than the hidden 23 us of k01_kernel.)
        rc = launch_k01_with_e1(m, p);  // (E1 first: e1_kernel, or the changed views in k01's own arguments)
        if (rc) return rc;
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
    ecc_stamp(m, 5);
    if (!pairs_launched) {
        if (ctx->timing) HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
        HIP_TRY(ecc_launch_pairs(&p, ctx->stream));
        if (ctx->timing) {
            HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
            ctx->ev_valid[0] = true;
        }
    }
    if (m->stamps[4] == 0.0) ecc_stamp(m, 4);
    if (sum_d) {
        if (count > 0) HIP_TRY(ecc_launch_sum_pairs(pair_values_d, count, sum_d, m->sum_scratch_d, ctx->stream));
        else if (sum_d == m->sum_h_dev) std::memset(m->sum_h, 0, sizeof(double));  // empty shard: nothing is launched
        else HIP_TRY(hipMemsetAsync(sum_d, 0, sizeof(double), ctx->stream));
    }
    m->rec_valid = m->record_reuse && !K01_d && count > 0;
    ecc_stamp(m, 6);
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
    ecc_mark_busy(m);  // (queues work; the synchronous callers set it again once they have seen the result)
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
                rc = fill_pair_params(m, &p, n * (n - 1) / 2, /*need_e1=*/false);  // the sampling mode of the full evaluation
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
                rc = launch_k01_with_e1(m, p);  // (the moved views' E1 in the launch's own arguments: no e1_kernel per step)
                if (rc) return rc;
                if (ctx->timing) HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
                HIP_TRY(ecc_launch_pairs(&p, ctx->stream));
                if (ctx->timing) {
                    HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
                    ctx->ev_valid[0] = true;
                }
            }
            if (sum_d) HIP_TRY(ecc_launch_sum_pairs(m->cache_values_d, count, sum_d, m->sum_scratch_d, ctx->stream));  // (null: the pose batch wants the values only)
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

}  // namespace ecc_internal

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

// ecc_rccl.cpp
struct ecc_comm;
extern "C" int ecc_comm_allreduce_sum_f64(ecc_comm* c, double* value_d);
extern "C" ecc_ctx* ecc_comm_context(ecc_comm* c);

// One rank's share of a sharded evaluation with the exchange inside the call: this rank's pairs -> float64 sum on the device
// -> RCCL all-reduce over the communicator's ranks -> the scalar published to the pinned result slot -> poll.  Four
// stream-ordered steps on the context's stream, one host call, nothing between them that waits for the host.
// ref: MetricRadonIntermediate::evaluate's sum (...RadonIntermediate.cpp:216-224), sharded.
ECC_EXPORT int ecc_metric_evaluate_range_allreduce(ecc_metric* m, ecc_comm* comm, int64_t first, int64_t count, double* sum_all)
{
    if (!m || !comm || !sum_all) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (ecc_comm_context(comm) != m->ctx) return fail(ECC_ERR_INVALID_ARGUMENT, "the communicator belongs to another context");
    ecc_stamp(m, 2);
    ecc_ctx* ctx = m->ctx;
    int rc = set_device(ctx);
    if (rc) return rc;
    rc = ensure_capacity(&m->pair_values_d, &m->pair_capacity, count > 0 ? count : 1, ctx->stream);
    if (rc) return rc;
    arm_result(m);
    // (the partial sum goes to the metric's device scalar, not to the result slot: the one-launch path for few pairs -- which
    // adds on the host -- does not apply, and a shard of an evaluation runs in the whole evaluation's sampling mode anyway)
    rc = launch_range(m, first, count, m->pair_values_d, nullptr, nullptr, m->sum_d, /*synchronous=*/true);
    if (rc) return rc;
    m->last_evaluated_pairs = count;
    rc = ecc_comm_allreduce_sum_f64(comm, m->sum_d);
    if (rc) return rc;
    HIP_TRY(ecc_launch_publish_scalar(m->sum_d, m->sum_h_dev, ctx->stream));
    HIP_TRY(wait_result(m, ctx->stream, sum_all));
    ecc_stamp(m, 7);
    m->done_generation = m->set_generation;
    m->quiet = true;  // the published scalar is the last thing this call queued, and it has been seen
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_publish_scalar(ecc_metric* m, const double* value_d)
{
    if (!m || !value_d) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    ecc_mark_busy(m);
    int rc = set_device(m->ctx);
    if (rc) return rc;
    arm_result(m);
    HIP_TRY(ecc_launch_publish_scalar(value_d, m->sum_h_dev, m->ctx->stream));
    m->publish_seq = m->queue_seq;  // nothing has been queued behind the publishing kernel yet
    m->publish_generation = m->set_generation;  // every launch so far read the staging buffers of generations up to this one
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_wait_scalar(ecc_metric* m, double* value)
{
    if (!m || !value) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    int rc = set_device(m->ctx);
    if (rc) return rc;
    HIP_TRY(wait_result(m, m->ctx->stream, value));
    // The publishing kernel is ordered behind everything the metric had queued when ecc_metric_publish_scalar was called,
    // and it has run.  Work queued SINCE (the next asynchronous range of a pipelined caller) may still be pending: only
    // when there is none is the metric quiet; the staging buffers known to be free are those of the set_projections calls
    // made before the publish.
    m->done_generation = std::max(m->done_generation, m->publish_generation);
    if (m->queue_seq == m->publish_seq) m->quiet = true;
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_evaluate_range(ecc_metric* m, int64_t first, int64_t count, float* pair_values,
                                         double* partial_sum)
{
    if (!m || !partial_sum) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    ecc_stamp(m, 2);
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
    ecc_stamp(m, 7);
    // an empty shard launches no kernel behind e1_kernel: its result slot says nothing about the stream
    if (count == 0) HIP_TRY(wait_stream_spin(ctx->stream));
    m->done_generation = m->set_generation;
    m->quiet = true;  // the result is the last thing this call queued, and it has been seen
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_evaluate_all(ecc_metric* m, float* cost_nxn, double* mean)
{
    if (!m || !mean) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    ecc_stamp(m, 2);
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
    ecc_stamp(m, 7);
    m->done_generation = m->set_generation;
    m->quiet = true;  // the result is the last thing this call queued, and it has been seen
    *mean = sum / (double)n_pairs;  // ref: ...RadonIntermediate.cpp:224 (all weights are 1)
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_evaluate_pairs(ecc_metric* m, const int32_t* idx4, int n_pairs, float* out, double* mean)
{
    if (m) ecc_mark_busy(m);
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
    rc = launch_k01_with_e1(m, p);
    if (rc) return rc;
    if (ctx->timing) HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
    HIP_TRY(ecc_launch_pairs(&p, ctx->stream));
    if (ctx->timing) {
        HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
        ctx->ev_valid[0] = true;
    }
    arm_result(m);
    if (pinned) {
        if (out) {
            arm_values(m, n_pairs);
            std::atomic_thread_fence(std::memory_order_seq_cst);
        }
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
        HIP_TRY(wait_values(m, n_pairs));
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
    if (m) ecc_mark_busy(m);
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
        *o++ = (float)(r.poly_ok & ~1) + ((r.poly_ok & 1) ? 0.5f : 0.f);  // the degree; + 0.5: the clamp-free class
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

// ---- experiment hooks (include/ecc_hip.h, last section): nothing of this kind is read from the environment ----------
ECC_EXPORT int ecc_debug_set_poly_tolerance(ecc_metric* m, float tol_bins)
{
    if (!m) return fail(ECC_ERR_INVALID_ARGUMENT, "metric is null");
    if (!(tol_bins >= 0.f) || tol_bins > 1.f) return fail(ECC_ERR_INVALID_ARGUMENT, "tolerance outside [0, 1] bins");
    m->economise_tol = tol_bins;
    m->rec_valid = false;    // the kept records and values were made with the old tolerance
    m->cache_valid = false;
    return ECC_OK;
}

ECC_EXPORT int ecc_debug_step_stamps(const ecc_metric* m, double* out8)
{
    if (!m || !out8) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    std::memcpy(out8, m->stamps, sizeof(double) * ECC_STEP_STAMPS);
    return ECC_OK;
}

ECC_EXPORT int ecc_debug_set_small_eval_bound(ecc_metric* m, int64_t max_pairs)
{
    if (!m) return fail(ECC_ERR_INVALID_ARGUMENT, "metric is null");
    m->small_max_pairs = max_pairs < 0 ? -1 : max_pairs;
    return ECC_OK;
}

/* Builds with -DECC_SMALL_STAMPS only: the wall-clock stamps (100 MHz) of the last small_eval_kernel launch, 4 per workgroup. */
ECC_EXPORT int ecc_debug_small_stamps(unsigned long long* out, int n_blocks)
{
    if (!g_small_dbg || !out || n_blocks < 1 || n_blocks > 4096) return ECC_ERR_INVALID_ARGUMENT;
    if (hipDeviceSynchronize() != hipSuccess) return ECC_ERR_HIP;
    return hipMemcpy(out, g_small_dbg, sizeof(unsigned long long) * 4 * n_blocks, hipMemcpyDeviceToHost) == hipSuccess ? ECC_OK : ECC_ERR_HIP;
}

