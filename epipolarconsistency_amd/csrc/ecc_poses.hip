// ecc_poses.hip -- many poses of ONE data set per call (BASELINE config 5: the 6-DoF sweep of one view; the probes of a
// finite-difference gradient).  ref: Gui/Visualization.h:59-112 (plotCostFunction: 100 steps x 6 parameters, one
// setProjectionMatrices + evaluate each), Gui/SingleImageMotion.h:84-90 (the objective those steps call).
//
// The reference -- and ecc_metric_set_projections + ecc_metric_evaluate_all here -- evaluates a pose at a time.  A pose that
// differs from a BASE set of matrices in one view changes n - 1 of the n (n - 1) / 2 pair values: ~1.5 us of pair-kernel work
// at 400 views inside a ~37-us step of launches, polls and hand-overs.  When the poses are known up front (a sweep, a
// gradient) nothing forces that cadence.  ecc_metric_evaluate_pose_deltas does, for K poses:
//   base values   the pair values of the base matrices (evaluate_cached: kept between calls, only what changed is redone;
//                 nothing is launched when nothing changed)
//   pose_list     ONE launch: the index grid -- entry (partner u, column q) = pair {u, moved view of column q}, the moved view's
//                 geometry taken from entry n + q of the EXTENDED arrays; a column per (pose, moved view) -- and, in the same
//                 launch, E1 of the base views followed by all moved matrices (the reference's Householder QR, bit-identical)
//   k01_kernel    ONE launch over the n x Q grid
//   pairs_kernel  ONE launch over it (partner-major: neighbours in the launch sample the same two Radon intermediates under
//                 slightly different geometries -- their lines are shared in the L1 / L2)
//   sum_poses     ONE launch: per pose, the float64 sum over ALL n (n - 1) / 2 values -- the base's with the pose's own
//                 substituted -- in exactly the order sum_pairs_kernel / sum_pairs_split_kernel add them
//   finish_poses  the slice sums of a pose added in slice order, stored to pinned host memory
// A pair value depends on its two matrices, its two Radon intermediates and the parameters only, the sampling mode is the one
// an evaluation of n (n - 1) / 2 pairs resolves to, and the sum's order is reproduced: every mean has the bits of
// ecc_metric_set_projections + ecc_metric_evaluate_all for that pose (tests/test_gpu_pose_batch.py).
#include "ecc_capi_internal.h"
#include "ecc_pose_diff.h"

using namespace ecc_internal;

#ifndef ECC_POSE_BATCH_MAX_MOVED
#define ECC_POSE_BATCH_MAX_MOVED 32  // moved views per pose the batch takes; a pose with more is evaluated the sequential way
#endif
#ifndef ECC_POSE_BATCH_MAX_ENTRIES
#define ECC_POSE_BATCH_MAX_ENTRIES (1 << 20)  // grid entries (records of 296 bytes) per batch: longer pose lists go in several batches
#endif

namespace {

constexpr int SUM_THREADS = 1024;   // sum_pairs_kernel / sum_pairs_split_kernel (pairs_kernel.hip): the order below is theirs
constexpr int SUM_SLICES = 16;      // SUM_BLOCKS of sum_pairs_split_kernel
constexpr int STAGE_F4 = 2 * SUM_THREADS;  // float4 per staged chunk: every thread's k, k + 1024 -- its own order is kept across chunks

struct PoseLists {
    const int32_t* off;    // n_poses + 1: first column of each pose (off[0] = 0, off[n_poses] = Q)
    const int32_t* views;  // Q moved views, ascending within a pose
};

// One workgroup per pose: the index tuples (P0, P1, dtr0, dtr1) of the pose's columns, entry (u, q) at u * Q + q.
//   partner u, moved view v = views[q]: the pair {min, max}; a view's geometry index is the view itself, or n + (its column)
//   when this pose moves it.  Holes -- u == v, and u < v when the pose moves u as well (that pair belongs to u's column) --
//   become (0, 0, 0, 0): a pair of a view with itself, whose value (0) nobody reads.
// Also copies the lists from the pinned block into device memory for sum_poses_kernel.
// Workgroups past the poses': E1 of the extended matrices (e1_kernel's work, geometry_kernel.hip -- the same code, ecc_host_geometry.h,
// the same bits; ref: ...RadonIntermediate.cpp:134-163): wave 0 of such a workgroup (P^+)^T of 64 views, wave 1 their source positions.
__global__ __launch_bounds__(256) void pose_list_kernel(PoseLists in, int n, int n_poses, int Q, int32_t* __restrict__ idx,
                                                        int32_t* __restrict__ lists_d, const double* __restrict__ Ps_ext,
                                                        float* __restrict__ PinvTs, float* __restrict__ Cs)
{
    __shared__ int M[ECC_POSE_BATCH_MAX_MOVED];
    if ((int)blockIdx.x >= n_poses) {  // uniform over the workgroup
        const int role = threadIdx.x >> 6;
        const int v = ((int)blockIdx.x - n_poses) * 64 + (threadIdx.x & 63);
        if (role > 1 || v >= n + Q) return;
        double P[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) P[q] = Ps_ext[12 * (size_t)v + q];
        if (role == 0) {
            float pinvT[12];
            ecc_host::pinv_transpose(P, pinvT);
#pragma unroll
            for (int q = 0; q < 12; ++q) PinvTs[12 * (size_t)v + q] = pinvT[q];
        } else {
            float C[4];
            ecc_host::source_position(P, C);
#pragma unroll
            for (int q = 0; q < 4; ++q) Cs[4 * (size_t)v + q] = C[q];
        }
        return;
    }
    const int k = blockIdx.x;
    const int o0 = in.off[k], c = in.off[k + 1] - o0;
    if ((int)threadIdx.x < c) {
        M[threadIdx.x] = in.views[o0 + threadIdx.x];
        lists_d[n_poses + 1 + o0 + threadIdx.x] = M[threadIdx.x];
    }
    if (threadIdx.x == 0) {
        lists_d[k] = o0;
        if (k == n_poses - 1) lists_d[n_poses] = o0 + c;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < c * n; e += 256) {
        const int a = e / n, u = e - a * n, v = M[a];
        bool hole = u == v;
        int pu = u;
        for (int b = 0; b < c; ++b)
            if (M[b] == u) {
                if (b < a) hole = true;
                else pu = n + o0 + b;
            }
        int4 t = make_int4(0, 0, 0, 0);
        if (!hole) {
            const int pv = n + o0 + a;
            t = u < v ? make_int4(pu, pv, u, v) : make_int4(pv, pu, v, u);
        }
        reinterpret_cast<int4*>(idx)[(size_t)u * Q + o0 + a] = t;
    }
}

// The float64 sum of a pose's `count` pair values = base[0 .. count) with the pose's own values substituted, in the order of
// sum_pairs_kernel (SLICES = 1: counts below 32 768) or sum_pairs_split_kernel (SLICES = 16) -- pairs_kernel.hip, ref:
// ...RadonIntermediate.cpp:216-224: slice s covers the float4 [s * per, min(n4, (s + 1) * per)); thread t of 1024 adds the
// components of its float4 k = lo + t, lo + t + 1024, ... into four accumulators, (a0 + a1) + (a2 + a3), thread 0 of the last
// slice then the up to three values past the last float4, the shuffle-down tree over a wave, the 16 wave sums in order.
// Workgroup (slice, pose): the slice is staged through LDS in chunks of 2048 float4, the pose's values that fall into the chunk
// are scattered over the base's, and the threads add from LDS.  IEEE binary64 additions in the same order: the same bits.
template <int SLICES>
__global__ __launch_bounds__(SUM_THREADS) void sum_poses_kernel(const float* __restrict__ base, long long count, int n, int Q,
                                                                const int32_t* __restrict__ lists_d, int n_poses,
                                                                const float* __restrict__ vals, double* __restrict__ partial,
                                                                double* __restrict__ out_host)
{
    __shared__ float stage[4 * STAGE_F4];
    __shared__ float tail[4];
    __shared__ int M[ECC_POSE_BATCH_MAX_MOVED];
    __shared__ double s[SUM_THREADS / 64];
    const int k = blockIdx.y, slice = blockIdx.x, t = threadIdx.x;
    const int o0 = lists_d[k], c = lists_d[k + 1] - o0;
    if (t < c) M[t] = lists_d[n_poses + 1 + o0 + t];
    const long long n4 = count >> 2;
    const long long per = (n4 + SLICES - 1) / SLICES;
    const long long lo = (long long)slice * per, hi = min(n4, lo + per);
    const bool owns_tail = slice == SLICES - 1;
    const float4* __restrict__ b4 = reinterpret_cast<const float4*>(base);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (owns_tail && t < 4) tail[t] = (n4 << 2) + t < count ? base[(n4 << 2) + t] : 0.f;
    bool first_chunk = true;
    for (long long c0 = lo; c0 < hi || first_chunk; c0 += STAGE_F4) {
        const long long ce = min(hi, c0 + STAGE_F4);
        for (long long kk = c0 + t; kk < ce; kk += SUM_THREADS) reinterpret_cast<float4*>(stage)[kk - c0] = b4[kk];
        __syncthreads();  // the chunk of the base (and M, tail) is in LDS
        // the pose's own values over the base's: column a of the pose, partner u -> pair {u, M[a]} at ij in the get_ij order
        for (int e = t; e < c * n; e += SUM_THREADS) {
            const int a = e / n, u = e - a * n, v = M[a];
            bool hole = u == v;
            for (int b = 0; b < a; ++b) hole = hole || M[b] == u;
            if (hole) continue;
            const long long i = u < v ? u : v, j = u < v ? v : u;
            const long long ij = i * n - i * (i + 1) / 2 + (j - i - 1);
            const bool in_chunk = ij >= (c0 << 2) && ij < (ce << 2);
            const bool in_tail = owns_tail && first_chunk && ij >= (n4 << 2);
            if (in_chunk || in_tail) {
                const float val = vals[(size_t)u * Q + o0 + a];
                if (in_chunk) stage[ij - (c0 << 2)] = val;
                else tail[ij - (n4 << 2)] = val;
            }
        }
        __syncthreads();
        for (long long kk = c0 + t; kk < ce; kk += SUM_THREADS) {
            const float4 v = reinterpret_cast<const float4*>(stage)[kk - c0];
            a0 += (double)v.x;
            a1 += (double)v.y;
            a2 += (double)v.z;
            a3 += (double)v.w;
        }
        __syncthreads();  // before the next chunk overwrites the stage
        first_chunk = false;
    }
    double acc = (a0 + a1) + (a2 + a3);
    if (owns_tail && t == 0)
        for (long long q = n4 << 2; q < count; ++q) acc += (double)tail[q - (n4 << 2)];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((t & 63) == 0) s[t >> 6] = acc;
    __syncthreads();
    if (t == 0) {
        double part = 0.0;
        for (int w = 0; w < SUM_THREADS / 64; w++) part += s[w];
        if (SLICES == 1)  // sum_pairs_kernel's own last step: this IS the result (no finish_poses_kernel launch)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(out_host) + k, (unsigned long long)__double_as_longlong(part),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else partial[(size_t)k * SLICES + slice] = part;
    }
}

// Pose k's slice sums in slice order (sum_pairs_split_kernel's last arriver; with one slice: 0.0 + the sum, the same number)
// -> the pinned result array (system-scope store: visible to the host before the stream is reported idle).
__global__ __launch_bounds__(256) void finish_poses_kernel(const double* __restrict__ partial, int slices, int n_poses,
                                                           double* __restrict__ out_host)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_poses) return;
    double tot = 0.0;
    for (int b = 0; b < slices; ++b) tot += partial[(size_t)k * slices + b];
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(out_host) + k, (unsigned long long)__double_as_longlong(tot),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

double clock_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
constexpr uint64_t POSE_PENDING = 0x7ff8ecc0dead0002ull;  // the two-deep form's "pending" pattern in the two result slots it uses

size_t align64(size_t x) { return (x + 63) & ~(size_t)63; }

int ensure_pose_block(ecc_metric* m, size_t bytes)
{
    if ((size_t)m->pose_h_bytes >= bytes) return ECC_OK;
    HIP_TRY(wait_stream_spin(m->ctx->stream));  // a launch may still be reading the old block
    if (m->pose_h) HIP_TRY(hipHostFree(m->pose_h));
    m->pose_h = m->pose_h_dev = nullptr;
    m->pose_h_bytes = 0;
    const size_t cap = std::max<size_t>(bytes + bytes / 2, 1 << 16);
    HIP_TRY(hipHostMalloc((void**)&m->pose_h, cap, hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void**)&m->pose_h_dev, m->pose_h, 0));
    m->pose_h_bytes = (int64_t)cap;
    return ECC_OK;
}

// One batch: poses with off[0] = 0 ... off[K] = Q columns, every pose at most ECC_POSE_BATCH_MAX_MOVED of them.
// base: the matrices the poses are deltas of (n x 12); base_vals_d: the n (n - 1) / 2 pair values of the base on the device.
// sums[k]: the float64 sum over all pair values of pose k.
int run_batch(ecc_metric* m, const double* base, const float* base_vals_d, int K, const int32_t* off, const int32_t* views,
              const double* moved_Ps, double* sums)
{
    ecc_ctx* ctx = m->ctx;
    const int64_t n = m->n_views, n_pairs = n * (n - 1) / 2;
    const int Q = off[K];
    const int64_t entries = n * (int64_t)Q;
    // the pinned block: extended matrices | results | off | views
    const size_t b_Ps = 0, b_out = align64(sizeof(double) * 12 * (size_t)(n + Q)), b_off = b_out + align64(sizeof(double) * (size_t)K),
                 b_views = b_off + align64(sizeof(int32_t) * (size_t)(K + 1)), b_end = b_views + align64(sizeof(int32_t) * (size_t)std::max(Q, 1));
    int rc = ensure_pose_block(m, b_end);
    if (rc) return rc;
    double* Ps_ext = reinterpret_cast<double*>(m->pose_h + b_Ps);
    volatile uint64_t* out = reinterpret_cast<volatile uint64_t*>(m->pose_h + b_out);
    std::memcpy(Ps_ext, base, sizeof(double) * 12 * (size_t)n);
    if (Q > 0) std::memcpy(Ps_ext + 12 * (size_t)n, moved_Ps, sizeof(double) * 12 * (size_t)Q);
    std::memcpy(m->pose_h + b_off, off, sizeof(int32_t) * (size_t)(K + 1));
    if (Q > 0) std::memcpy(m->pose_h + b_views, views, sizeof(int32_t) * (size_t)Q);
    for (int k = 0; k < K; ++k) out[k] = POSE_PENDING;
    std::atomic_thread_fence(std::memory_order_seq_cst);

    rc = ensure_capacity(&m->pose_PinvTs_d, &m->pose_PinvTs_capacity, 12 * (n + Q), ctx->stream);
    if (!rc) rc = ensure_capacity(&m->pose_Cs_d, &m->pose_Cs_capacity, 4 * (n + Q), ctx->stream);
    if (!rc) rc = ensure_capacity(&m->pose_idx_d, &m->pose_idx_capacity, 4 * std::max<int64_t>(entries, 1), ctx->stream);
    if (!rc) rc = ensure_capacity(&m->pose_records_d, &m->pose_records_capacity, std::max<int64_t>(entries, 1), ctx->stream);
    if (!rc) rc = ensure_capacity(&m->pose_values_d, &m->pose_values_capacity, std::max<int64_t>(entries, 1), ctx->stream);
    if (!rc) rc = ensure_capacity(&m->pose_partial_d, &m->pose_partial_capacity, (int64_t)K * SUM_SLICES, ctx->stream);
    if (!rc) rc = ensure_capacity(&m->pose_lists_d, &m->pose_lists_capacity, (int64_t)K + 1 + std::max(Q, 1), ctx->stream);
    if (rc) return rc;

    EccPairParams p;
    rc = fill_pair_params(m, &p, n_pairs, /*need_e1=*/false);  // the sampling mode of an all-pairs evaluation
    if (rc) return rc;
    PoseLists in = {reinterpret_cast<const int32_t*>(m->pose_h_dev + b_off), reinterpret_cast<const int32_t*>(m->pose_h_dev + b_views)};
    // index tuples + lists (K workgroups) and E1 of the n + Q extended matrices (the workgroups behind them): one launch
    const unsigned e1_blocks = entries > 0 ? (unsigned)((n + Q + 63) / 64) : 0u;
    hipLaunchKernelGGL(pose_list_kernel, dim3((unsigned)K + e1_blocks), dim3(256), 0, ctx->stream, in, (int)n, K, Q, m->pose_idx_d,
                       m->pose_lists_d, reinterpret_cast<const double*>(m->pose_h_dev + b_Ps), m->pose_PinvTs_d, m->pose_Cs_d);
    HIP_TRY(hipGetLastError());
    if (entries > 0) {
        p.PinvTs = m->pose_PinvTs_d;
        p.Cs = m->pose_Cs_d;
        p.indices = m->pose_idx_d;
        p.records = m->pose_records_d;
        p.pair_values = m->pose_values_d;
        p.first = 0;
        p.count = entries;
        HIP_TRY(ecc_launch_k01(&p, ctx->stream));
        if (ctx->timing) HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
        HIP_TRY(ecc_launch_pairs(&p, ctx->stream));
        if (ctx->timing) {
            HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
            ctx->ev_valid[0] = true;
        }
    }
    const int slices = n_pairs >= 32768 ? SUM_SLICES : 1;  // ecc_launch_sum_pairs (pairs_kernel.hip)
    double* out_dev = reinterpret_cast<double*>(m->pose_h_dev + b_out);
    if (slices == 1)
        hipLaunchKernelGGL(sum_poses_kernel<1>, dim3(1, (unsigned)K), dim3(SUM_THREADS), 0, ctx->stream, base_vals_d, (long long)n_pairs,
                           (int)n, Q, m->pose_lists_d, K, m->pose_values_d, m->pose_partial_d, out_dev);
    else
        hipLaunchKernelGGL(sum_poses_kernel<SUM_SLICES>, dim3(SUM_SLICES, (unsigned)K), dim3(SUM_THREADS), 0, ctx->stream, base_vals_d,
                           (long long)n_pairs, (int)n, Q, m->pose_lists_d, K, m->pose_values_d, m->pose_partial_d, out_dev);
    HIP_TRY(hipGetLastError());
    if (slices > 1) {
        hipLaunchKernelGGL(finish_poses_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, ctx->stream, m->pose_partial_d, slices, K, out_dev);
        HIP_TRY(hipGetLastError());
    }
    // the results arrive in pinned memory a few microseconds before the stream is reported idle: poll the last one, then the rest
    double t0 = 0.0;
    for (unsigned spins = 0;; ++spins) {
        int k = K - 1;
        while (k >= 0 && out[k] != POSE_PENDING) --k;
        if (k < 0) break;
        if ((spins & 0xfff) == 0xfff) {
            const double t = clock_now();
            if (t0 == 0.0) t0 = t;
            else if (t - t0 > 2.0) {
                HIP_TRY(hipStreamSynchronize(ctx->stream));
                for (k = 0; k < K; ++k)
                    if (out[k] == POSE_PENDING) return fail(ECC_ERR_HIP, "the pose batch ran and stored no result");
                break;
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    for (int k = 0; k < K; ++k) {
        const uint64_t bits = out[k];
        std::memcpy(&sums[k], &bits, sizeof(double));
    }
    return ECC_OK;
}

// The poses first, first + stride, ... of Ps_batch one after the other on the context's stream, two deep: pose k + 1's
// hand-over (staging of the matrices, the comparison with the kept records, host E1 of the changed views, the launches) is done
// while the device still runs pose k, whose result is polled only afterwards -- the ~20 us between a result and the next
// evaluation's first kernel (poll, caller, hand-over, dispatch) disappear from all evaluations but the first.  The device
// executes exactly the launches of ecc_metric_set_projections + ecc_metric_evaluate_all per pose, in the same order.
int poses_two_deep(ecc_metric* m, const std::vector<int>& which, const double* Ps_batch, int n_views, double* means)
{
    ecc_ctx* ctx = m->ctx;
    const int64_t n = n_views, n_pairs = n * (n - 1) / 2;
    const int count = (int)which.size();
    if (count == 0) return ECC_OK;
    int rc = ECC_OK;
    // one at a time: the pose-delta mode (it keeps the values of ONE previous evaluation), and evaluations small enough
    // for the one-launch path (its hand-over goes through result slot 0 and the host's sum; a few microseconds of device
    // work leave nothing to overlap anyway)
    if (m->incremental || !m->sum_h || n_pairs <= ECC_SMALL_EVAL_MAX_PAIRS) {
        for (int q = 0; q < count; ++q) {
            rc = ecc_metric_set_projections(m, Ps_batch + (size_t)12 * n * which[q], n_views);
            if (rc) return rc;
            rc = ecc_metric_evaluate_all(m, nullptr, &means[which[q]]);
            if (rc) return rc;
        }
        return ECC_OK;
    }
    rc = ensure_capacity(&m->pair_values_d, &m->pair_capacity, n_pairs, ctx->stream);
    if (rc) return rc;
    volatile uint64_t* slots = reinterpret_cast<volatile uint64_t*>(m->sum_h);
    uint64_t gen_of[2] = {0, 0};
    auto collect = [&](int q) -> int {  // waits for the q-th pose's sum in slot q & 1 (bounded spin, then the stream)
        const int sl = q & 1;
        double t0 = 0.0;
        for (unsigned spins = 0;; ++spins) {
            const uint64_t bits = slots[sl];
            if (bits != POSE_PENDING) {
                double v;
                std::memcpy(&v, &bits, sizeof(v));
                means[which[q]] = v / (double)n_pairs;  // ref: ...RadonIntermediate.cpp:224 (all weights are 1)
                m->done_generation = gen_of[sl];
                return ECC_OK;
            }
            if ((spins & 0xfff) == 0xfff) {
                const double t = clock_now();
                if (t0 == 0.0) t0 = t;
                else if (t - t0 > 2.0) {
                    HIP_TRY(hipStreamSynchronize(ctx->stream));
                    if (slots[sl] == POSE_PENDING) return fail(ECC_ERR_HIP, "an evaluation ran and stored no result");
                }
            }
        }
    };
    for (int q = 0; q < count; ++q) {
        rc = ecc_metric_set_projections(m, Ps_batch + (size_t)12 * n * which[q], n_views);  // buffer q & 1: its last device reader was pose q - 2
        if (rc) return rc;
        const int sl = q & 1;
        slots[sl] = POSE_PENDING;
        std::atomic_thread_fence(std::memory_order_seq_cst);
        gen_of[sl] = m->set_generation;
        // (slot 1 is not the result slot of the synchronous calls: those launches never take the one-launch path)
        rc = launch_range(m, 0, n_pairs, m->pair_values_d, nullptr, nullptr, m->sum_h_dev + sl, /*synchronous=*/false);
        if (rc) return rc;
        m->small_pending_count = 0;
        if (q >= 1) {
            rc = collect(q - 1);
            if (rc) return rc;
        }
    }
    rc = collect(count - 1);
    if (rc) return rc;
    m->last_evaluated_pairs = n_pairs;
    return ECC_OK;
}

// The automatic object radius follows the FIRST matrix (ref: Metric::getObjectRadius, EpipolarConsistency.cpp:76-84): a pose
// that moves view 0 may change it, and with it every pair's record.  Such a pose is not a delta of the base.
bool pose_keeps_radius(const ecc_metric* m, double base_radius, int c, const int32_t* views, const double* moved_Ps)
{
    if (m->object_radius_mm > 0 || c < 1 || views[0] != 0) return true;
    // (the launches take the radius as a float: fill_pair_params)
    return (float)ecc_host::object_radius(moved_Ps, m->n_u, m->n_v) == (float)base_radius;
}

// The batched poses of a call: lists relative to the metric's CURRENT matrices.  sums -> means.
int evaluate_deltas(ecc_metric* m, int n_poses, const int32_t* off, const int32_t* views, const double* moved_Ps, double* means,
                    std::vector<int>* not_batched)
{
    ecc_ctx* ctx = m->ctx;
    const int64_t n = m->n_views, n_pairs = n * (n - 1) / 2;
    ecc_mark_busy(m);
    std::vector<double> base(m->Ps_h[m->set_generation & 1], m->Ps_h[m->set_generation & 1] + 12 * n);
    double base_radius = 0.0;
    ecc_metric_get_object_radius(m, &base_radius);
    // the base's pair values: kept between calls (the pose-delta cache), only the pairs of views that changed since are redone
    float* base_vals_d = nullptr;
    int rc = evaluate_cached(m, 0, n_pairs, /*sum_d=*/nullptr, &base_vals_d);  // (the values only: the segmented sum adds them per pose)
    if (rc) return rc;
    std::vector<int32_t> b_off, b_views;
    std::vector<double> b_Ps, sums;
    std::vector<int> b_pose;
    const int64_t max_cols = std::max<int64_t>(ECC_POSE_BATCH_MAX_ENTRIES / n, ECC_POSE_BATCH_MAX_MOVED);
    auto flush = [&]() -> int {
        if (b_pose.empty()) return ECC_OK;
        sums.resize(b_pose.size());
        const int e = run_batch(m, base.data(), base_vals_d, (int)b_pose.size(), b_off.data(), b_views.data(), b_Ps.data(), sums.data());
        if (e) return e;
        for (size_t q = 0; q < b_pose.size(); ++q) means[b_pose[q]] = sums[q] / (double)n_pairs;  // ref: ...RadonIntermediate.cpp:224
        m->last_batched_poses += (int64_t)b_pose.size();
        b_pose.clear();
        b_off.assign(1, 0);
        b_views.clear();
        b_Ps.clear();
        return ECC_OK;
    };
    b_off.assign(1, 0);
    for (int k = 0; k < n_poses; ++k) {
        const int c = off[k + 1] - off[k];
        const int32_t* vk = views + off[k];
        const double* Pk = moved_Ps + 12 * (size_t)off[k];
        if (c > ECC_POSE_BATCH_MAX_MOVED || !pose_keeps_radius(m, base_radius, c, vk, Pk)) {
            not_batched->push_back(k);
            continue;
        }
        if ((int64_t)b_views.size() + c > max_cols || b_pose.size() >= 32768) {  // (the sum's grid is slices x poses: y < 65 536)
            rc = flush();
            if (rc) return rc;
        }
        b_pose.push_back(k);
        b_views.insert(b_views.end(), vk, vk + c);
        b_Ps.insert(b_Ps.end(), Pk, Pk + 12 * (size_t)c);
        b_off.push_back((int32_t)b_views.size());
    }
    rc = flush();
    if (rc) return rc;
    HIP_TRY(wait_stream_spin(ctx->stream));  // (the results were seen before the stream's own completion; the scratch is reused)
    m->quiet = true;
    return ECC_OK;
}

int check_lists(const ecc_metric* m, int n_poses, const int32_t* off, const int32_t* views)
{
    if (off[0] != 0) return fail(ECC_ERR_INVALID_ARGUMENT, "moved_offsets[0] must be 0");
    for (int k = 0; k < n_poses; ++k) {
        if (off[k + 1] < off[k]) return fail(ECC_ERR_INVALID_ARGUMENT, "moved_offsets must not decrease");
        for (int q = off[k]; q < off[k + 1]; ++q) {
            if (views[q] < 0 || views[q] >= m->n_views) return fail(ECC_ERR_INVALID_ARGUMENT, "moved view outside [0, n_views)");
            if (q > off[k] && views[q] <= views[q - 1])
                return fail(ECC_ERR_INVALID_ARGUMENT, "the moved views of a pose must be strictly ascending");
        }
    }
    return ECC_OK;
}

}  // namespace

ECC_EXPORT int ecc_metric_set_pose_batching(ecc_metric* m, int on)
{
    if (!m) return fail(ECC_ERR_INVALID_ARGUMENT, "metric is null");
    m->pose_batching = on ? 1 : 0;
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_last_batched_poses(const ecc_metric* m, int64_t* poses)
{
    if (!m || !poses) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    *poses = m->last_batched_poses;
    return ECC_OK;
}

ECC_EXPORT int ecc_metric_evaluate_pose_deltas(ecc_metric* m, int n_poses, const int32_t* moved_offsets, const int32_t* moved_views,
                                               const double* moved_Ps, double* means)
{
    if (!m || !moved_offsets || !means) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (n_poses < 1) return ECC_OK;
    if (moved_offsets[n_poses] > 0 && (!moved_views || !moved_Ps)) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (m->n_views < 2) return fail(ECC_ERR_INVALID_ARGUMENT, "need at least two views (the reference divides 0/0 here)");
    if ((int)m->dtrs.size() < m->n_views) return fail(ECC_ERR_INVALID_ARGUMENT, "fewer Radon intermediates than projection matrices");
    int rc = set_device(m->ctx);
    if (rc) return rc;
    rc = check_lists(m, n_poses, moved_offsets, moved_views);
    if (rc) return rc;
    m->last_batched_poses = 0;
    const int64_t n = m->n_views;
    std::vector<int> rest;
    if (m->pose_batching) {
        rc = evaluate_deltas(m, n_poses, moved_offsets, moved_views, moved_Ps, means, &rest);
        if (rc) return rc;
    } else {
        for (int k = 0; k < n_poses; ++k) rest.push_back(k);
    }
    if (rest.empty()) return ECC_OK;
    // what the batch does not take (more moved views than it handles, a changed automatic radius, batching off): the pose's
    // full matrices the sequential way, then the base again
    const std::vector<double> base(m->Ps_h[m->set_generation & 1], m->Ps_h[m->set_generation & 1] + 12 * n);
    std::vector<double> full(base);
    for (int k : rest) {
        for (int q = moved_offsets[k]; q < moved_offsets[k + 1]; ++q)
            std::memcpy(full.data() + 12 * (size_t)moved_views[q], moved_Ps + 12 * (size_t)q, sizeof(double) * 12);
        rc = ecc_metric_set_projections(m, full.data(), (int)n);
        if (!rc) rc = ecc_metric_evaluate_all(m, nullptr, &means[k]);
        if (rc) break;
        for (int q = moved_offsets[k]; q < moved_offsets[k + 1]; ++q)
            std::memcpy(full.data() + 12 * (size_t)moved_views[q], base.data() + 12 * (size_t)moved_views[q], sizeof(double) * 12);
    }
    const int rb = ecc_metric_set_projections(m, base.data(), (int)n);
    return rc ? rc : rb;
}

// ref for the pattern: Gui/Visualization.h:78-98 plotCostFunction, BASELINE config 5; a finite-difference gradient.
ECC_EXPORT int ecc_metric_evaluate_poses_strided(ecc_metric* m, int n_poses, const double* Ps_batch, int n_views, int first, int stride,
                                                 double* means)
{
    if (m) ecc_mark_busy(m);
    if (!m || !Ps_batch || !means) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (first < 0 || stride < 1) return fail(ECC_ERR_INVALID_ARGUMENT, "first must be >= 0 and stride >= 1");
    if (n_poses < 1 || first >= n_poses) return ECC_OK;
    if (n_views < 2) return fail(ECC_ERR_INVALID_ARGUMENT, "need at least two views (the reference divides 0/0 here)");
    int rc = set_device(m->ctx);
    if (rc) return rc;
    m->last_batched_poses = 0;
    const int64_t n = n_views;
    const size_t pose_doubles = 12 * (size_t)n;
    std::vector<int> mine;
    for (int p = first; p < n_poses; p += stride) mine.push_back(p);
    const int last = mine.back();
    if (!m->pose_batching || mine.size() < 2) return poses_two_deep(m, mine, Ps_batch, n_views, means);

    // The poses as deltas of a base: the metric's current matrices if at least half of the poses differ from them in a few
    // views, else the first pose (a sweep around an estimate the metric has not seen yet).
    // (the comparison: ecc_pose_diff.h, up to eight host threads)
    ecc_pose_diff::Result d;
    std::vector<int32_t>&off = d.off, &views = d.views;
    std::vector<int>&batch_pose = d.batch_pose, &rest = d.rest;
    auto diff_against = [&](const double* base) { ecc_pose_diff::diff(Ps_batch, n_views, mine, base, ECC_POSE_BATCH_MAX_MOVED, 8, &d); };
    bool have_base = m->n_views == n_views && m->set_generation > 0 && (int)m->dtrs.size() >= n_views;
    if (have_base) diff_against(m->Ps_h[m->set_generation & 1]);
    if (!have_base || 2 * batch_pose.size() < mine.size()) {
        if ((int)m->dtrs.size() < n_views) return fail(ECC_ERR_INVALID_ARGUMENT, "fewer Radon intermediates than projection matrices");
        diff_against(Ps_batch + pose_doubles * (size_t)mine[0]);
        if (2 * batch_pose.size() < mine.size()) return poses_two_deep(m, mine, Ps_batch, n_views, means);
        rc = ecc_metric_set_projections(m, Ps_batch + pose_doubles * (size_t)mine[0], n_views);
        if (rc) return rc;
    }
    std::vector<double> moved_Ps(12 * views.size());
    std::vector<double> batch_means(batch_pose.size());
    for (size_t b = 0; b < batch_pose.size(); ++b)
        for (int q = off[b]; q < off[b + 1]; ++q)
            std::memcpy(moved_Ps.data() + 12 * (size_t)q, Ps_batch + pose_doubles * (size_t)batch_pose[b] + 12 * (size_t)views[q], sizeof(double) * 12);
    std::vector<int> unbatched;
    rc = evaluate_deltas(m, (int)batch_pose.size(), off.data(), views.data(), moved_Ps.data(), batch_means.data(), &unbatched);
    if (rc) return rc;
    for (size_t b = 0; b < batch_pose.size(); ++b) means[batch_pose[b]] = batch_means[b];
    for (int b : unbatched) rest.push_back(batch_pose[b]);  // (a changed automatic radius)
    std::sort(rest.begin(), rest.end());
    rc = poses_two_deep(m, rest, Ps_batch, n_views, means);
    if (rc) return rc;
    // "the matrices of the last pose stay the metric's current ones" (include/ecc_hip.h)
    if (rest.empty() || rest.back() != last) rc = ecc_metric_set_projections(m, Ps_batch + pose_doubles * (size_t)last, n_views);
    return rc;
}

ECC_EXPORT int ecc_metric_evaluate_poses(ecc_metric* m, int n_poses, const double* Ps_batch, int n_views, double* means)
{
    return ecc_metric_evaluate_poses_strided(m, n_poses, Ps_batch, n_views, 0, 1, means);
}
