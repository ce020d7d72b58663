// ecc_capi_internal.h -- what the translation units of the C ABI share (ecc_capi.hip: errors, contexts; ecc_radon_api.hip: Radon
// intermediates; ecc_metric_api.hip: the metric's objects; ecc_evaluate.hip: its evaluation paths; ecc_preprocess_api.hip;
// ecc_direct_api.hip: MetricDirect): the object
// layouts behind the opaque handles, the kernel launchers, error handling and the helpers of namespace ecc_internal.
#ifndef ECC_CAPI_INTERNAL_H
#define ECC_CAPI_INTERNAL_H

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
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
extern "C" hipError_t ecc_launch_k01_patched(const EccPairParams* p, const EccSmallEval* x, hipStream_t stream);
extern "C" hipError_t ecc_launch_pairs(const EccPairParams* p, hipStream_t stream);
extern "C" int ecc_small_eval_plan(const EccPairParams* p, long long forced_bound, int* wpp, size_t* lds_bytes);
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

// Thread-local message of the last failing call (ecc_last_error); defined in ecc_capi.hip.
extern "C" int ecc_set_error(int code, const char* msg);
namespace {
inline int fail(int code, const std::string& msg) { return ecc_set_error(code, msg.c_str()); }
}  // namespace

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
struct EccSlab {
    float* ptr = nullptr;
    int device = 0;
    ~EccSlab()
    {
        if (ptr) {
            (void)hipSetDevice(device);
            (void)hipFree(ptr);
        }
    }
};
typedef EccSlab Slab;

struct ecc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool timing = false;
    int radon_arithmetic = ECC_RADON_EXACT;  // ecc_radon_set_arithmetic
    int quad_copies = ECC_QUAD_COPIES_AUTO;  // ecc_ctx_set_quad_copies: whether metrics created from this context build row-quad copies
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
    std::vector<const float*> quads_table_h; // host image of quads_table_d
    bool quads_possible = false;             // offsets into a row-quad copy fit 32 bits
    bool quads_decided = true;               // ECC_QUAD_COPIES_AUTO: false until the first large evaluation has looked at the matrices
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
    // automatic object radius (a function of the first matrix and the image size), kept until the first matrix changes
    mutable double radius_cache = 0.0;
    mutable double radius_cache_P[12] = {0};
    mutable bool radius_cache_valid = false;
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
    std::vector<int> scratch_stale;     // small_eval_patches: views whose geometry on the device is behind
    std::vector<int32_t> scratch_refs;
    std::vector<int32_t> scratch_patch_of;
    // second stream of the reuse path: refit + list launch of the changed pairs run there while the all-pairs launch
    // (which skips them) already runs on the context's stream
    hipStream_t side_stream = nullptr;
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    // Everything this metric has queued on the context's stream or its side stream is KNOWN to have completed: set by the
    // synchronous evaluations once they have seen their result (the last thing they queued), cleared by whatever queues work for
    // the metric.  The two-stream refit then needs no fork event: what the side stream reads -- records, device geometry, the
    // row-paired copies, the pinned lists -- is written by this metric's own launches only, so nothing it must wait for is pending.
    // (hipEventRecord + hipStreamWaitEvent in front of the first launch cost the shard step of an 8-rank job 3.5-4.7 us of 72.)
    bool quiet = false;
    // What `quiet` may be set from: every call that queues work for the metric bumps queue_seq (ecc_mark_busy);
    // ecc_metric_publish_scalar remembers the value right after its own kernel.  ecc_metric_wait_scalar has only seen THAT kernel
    // complete -- in a pipelined order (async k, publish k, async k + 1, wait k) later work is still pending, and the wait
    // must not declare the metric quiet (advisor, round 5).
    uint64_t queue_seq = 0, publish_seq = 0, publish_generation = 0;
    // ecc_metric_set_small_eval: evaluations of at most ECC_SMALL_EVAL_MAX_PAIRS pairs as ONE launch (small_eval_kernel.hip).
    // E1 of the views whose matrix changed since the device arrays were made is done on the host and handed over in the
    // kernel arguments (dev_Ps above says which views those are).
    int small_eval = 1;
    int64_t small_max_pairs = -1;  // ecc_debug_set_small_eval_bound: >= 0 replaces the one-launch path's own size bound
    float economise_tol = ECC_POLY_ECONOMISE_TOL_BINS;  // ecc_debug_set_poly_tolerance
    int32_t* sidx_h = nullptr;    // index list of a fused index-list evaluation (pinned, device-mapped)
    int32_t* sidx_h_dev = nullptr;
    int64_t sidx_capacity = 0;    // pairs
    float* svals_h = nullptr;     // pair values a caller wants on the host (pinned, device-mapped)
    float* svals_h_dev = nullptr;
    int64_t svals_capacity = 0;
    unsigned* small_ticket_d = nullptr;
    int64_t small_pending_count = 0;  // > 0: the result slot will receive the "done" word of a one-launch evaluation of that many pairs
    // ecc_metric_evaluate_pose_deltas (ecc_poses.hip): K poses as ONE record launch, ONE pair launch and ONE segmented sum.
    // Everything below is scratch of that path, grown on demand: the extended geometry (the base views, then one entry per
    // moved view of every pose), the pose-major x partner-major index grid, its records and values, the slice sums, and one
    // pinned, device-mapped block (extended matrices, the lists, the results).
    int pose_batching = 1;  // ecc_metric_set_pose_batching
    char* pose_h = nullptr;
    char* pose_h_dev = nullptr;
    int64_t pose_h_bytes = 0;
    float* pose_PinvTs_d = nullptr;
    int64_t pose_PinvTs_capacity = 0;  // floats
    float* pose_Cs_d = nullptr;
    int64_t pose_Cs_capacity = 0;      // floats
    int32_t* pose_idx_d = nullptr;
    int64_t pose_idx_capacity = 0;     // ints
    EccPairRecord* pose_records_d = nullptr;
    int64_t pose_records_capacity = 0;
    float* pose_values_d = nullptr;
    int64_t pose_values_capacity = 0;
    double* pose_partial_d = nullptr;
    int64_t pose_partial_capacity = 0;
    int32_t* pose_lists_d = nullptr;
    int64_t pose_lists_capacity = 0;   // ints
    int64_t last_batched_poses = 0;    // poses the last ecc_metric_evaluate_poses* call took through the batch (ecc_metric_last_batched_poses)
    // ecc_debug_step_stamps: host clock (seconds, steady) at fixed points of the last set_projections / synchronous evaluation
    double stamps[ECC_STEP_STAMPS] = {0};
};

inline void ecc_mark_busy(ecc_metric* m)
{
    m->quiet = false;
    ++m->queue_seq;
}

inline void ecc_stamp(ecc_metric* m, int k)
{
    m->stamps[k] = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

namespace ecc_internal {

hipError_t wait_stream_spin(hipStream_t stream);
void arm_result(ecc_metric* m);
hipError_t wait_result(ecc_metric* m, hipStream_t stream, double* value);
int set_device(const ecc_ctx* ctx);
int ensure_poly_tables(ecc_ctx* ctx);
void decide_quad_copies(ecc_metric* m);  // ECC_QUAD_COPIES_AUTO: build the row-quad copies if the current matrices' pairs would read them
int ensure_e1(ecc_metric* m);  // E1 on the device for the staged matrices, if the device arrays are behind them
// ecc_evaluate.hip: the parameters of a launch; the stream-ordered launches over a pair range; the pose-delta path
int fill_pair_params(ecc_metric* m, EccPairParams* p, int64_t mode_count, bool need_e1 = true);
int launch_range(ecc_metric* m, int64_t first, int64_t count, float* pair_values_d, float* cost_d, float* K01_d, double* sum_d,
                 bool synchronous = false);
int evaluate_cached(ecc_metric* m, int64_t first, int64_t count, double* sum_d, float** vals_out);

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

}  // namespace ecc_internal

#endif
