// ecc_group.cpp -- single-process multi-GPU form of the metric (host code only, built on the public C ABI).
//
// The reference is ONE process whose optimiser thread calls
//     ecc->setProjectionMatrices(Ps); double cost = ecc->evaluate();
// (ref: LibEpipolarConsistency/Gui/SingleImageMotion.h:84-90 under HeaderOnly/LibOpterix/WrapNLOpt.hxx:151-173,
// host epilogue ...RadonIntermediate.cpp:166-225).  A group gives that caller all GPUs of the node behind the same two
// calls: one context + stream + host thread per device, the Radon-intermediate stack replicated on every device
// (944 MB of 288 GB each), the pair range cut into contiguous equal-count shards of the get_ij order
// (ref: EpipolarConsistencyCommon.hxx:52-79) and the partial sums -- 8 bytes per device, already in pinned host memory --
// added on the host in rank order.  There is no device-side exchange on the per-evaluation path and no RCCL: the only
// "collective" of the path is that sum, and inside one process it is a loop over 8 doubles.
//
// Threads: rank 0 runs on the calling thread, ranks 1.. on persistent workers that spin briefly for the next job and
// then sleep on a condition variable, so an idle group costs no CPU.  Launching the shards of 8 devices from ONE thread
// would serialise ~4 launches x 8 devices of host time (~100 us) in front of a ~50 us shard.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ecc_hip.h"
#include "ecc_worker_pool.h"

#define ECC_EXPORT extern "C" __attribute__((visibility("default")))

extern "C" int ecc_set_error(int code, const char* msg);  // ecc_capi.hip: records the message for ecc_last_error()

struct ecc_group {
    std::vector<int> devices;
    std::vector<ecc_ctx*> ctxs;
    std::vector<hipStream_t> streams;
    EccWorkerPool pool;  // job hand-off to the workers of ranks 1..n-1 (ecc_worker_pool.h)

    int size() const { return (int)devices.size(); }
};

namespace {
std::atomic<int> g_force_replica{0};
}

/* Debug / test hook (not an environment variable: nothing a deployment can set by accident): the ranks of groups created
 * from now on copy the Radon-intermediate stack even when it already lives on their device. */
extern "C" __attribute__((visibility("default"))) int ecc_group_debug_force_replica(int on)
{
    g_force_replica.store(on ? 1 : 0);
    return 0;
}

namespace {

// Runs job(rank) for every rank (rank 0 on the calling thread) and returns the first failure, its message recorded
// for ecc_last_error() on the calling thread.
int run_all(ecc_group* g, std::function<int(int)> job)
{
    const int bad = g->pool.run_all(std::move(job), ECC_ERR_HIP);
    (void)hipSetDevice(g->devices[0]);
    if (bad >= 0) {
        const std::string msg = "rank " + std::to_string(bad) + " (device " + std::to_string(g->devices[bad]) + "): " + g->pool.message(bad);
        return ecc_set_error(g->pool.status(bad), msg.c_str());
    }
    return ECC_OK;
}

}  // namespace

ECC_EXPORT void ecc_pair_shard(int64_t n_pairs, int world, int rank, int64_t* first, int64_t* count)
{
    // contiguous, equal-count (+-1) chunks of the get_ij order; the same rule as sharding.pair_range (Python side)
    const int64_t a = (int64_t)rank * n_pairs / world, b = (int64_t)(rank + 1) * n_pairs / world;
    if (first) *first = a;
    if (count) *count = b - a;
}

ECC_EXPORT int ecc_group_create(int n_dev, const int* devices, ecc_group** out)
{
    if (!out) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "out is null");
    if (n_dev < 1 || n_dev > 64) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "group size must be in [1, 64]");
    const int visible = ecc_device_count();
    if (visible <= 0) return ecc_set_error(ECC_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    std::unique_ptr<ecc_group> g(new (std::nothrow) ecc_group());
    if (!g) return ecc_set_error(ECC_ERR_OUT_OF_MEMORY, "host allocation failed");
    for (int r = 0; r < n_dev; ++r) {
        const int d = devices ? devices[r] : r;
        if (d < 0 || d >= visible) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "device index out of range");
        g->devices.push_back(d);
    }
    int rc = ECC_OK;
    for (int r = 0; r < n_dev && rc == ECC_OK; ++r) {
        hipStream_t s = nullptr;
        if (hipSetDevice(g->devices[r]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            rc = ecc_set_error(ECC_ERR_HIP, "could not create a stream on a device of the group");
            break;
        }
        g->streams.push_back(s);
        ecc_ctx* c = nullptr;
        rc = ecc_ctx_create(g->devices[r], s, &c);
        if (rc == ECC_OK) g->ctxs.push_back(c);
    }
    // peer access where the hardware offers it (xGMI): the replication copies then go device to device
    for (int a = 0; a < n_dev && rc == ECC_OK; ++a)
        for (int b = 0; b < n_dev; ++b) {
            if (g->devices[a] == g->devices[b]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, g->devices[a], g->devices[b]) == hipSuccess && can) {
                (void)hipSetDevice(g->devices[a]);
                (void)hipDeviceEnablePeerAccess(g->devices[b], 0);  // "already enabled" is fine
            }
            (void)hipGetLastError();
        }
    if (rc != ECC_OK) {
        for (ecc_ctx* c : g->ctxs) ecc_ctx_destroy(c);
        for (size_t r = 0; r < g->streams.size(); ++r) {
            (void)hipSetDevice(g->devices[r]);
            (void)hipStreamDestroy(g->streams[r]);
        }
        return rc;
    }
    ecc_group* raw = g.release();
    raw->pool.start(n_dev, [raw](int r) { (void)hipSetDevice(raw->devices[r]); }, [] { return std::string(ecc_last_error()); });
    (void)hipSetDevice(raw->devices[0]);
    *out = raw;
    return ECC_OK;
}

ECC_EXPORT int ecc_group_destroy(ecc_group* g)
{
    if (!g) return ECC_OK;
    g->pool.stop();
    for (ecc_ctx* c : g->ctxs) ecc_ctx_destroy(c);
    for (size_t r = 0; r < g->streams.size(); ++r) {
        (void)hipSetDevice(g->devices[r]);
        (void)hipStreamDestroy(g->streams[r]);
    }
    delete g;
    return ECC_OK;
}

ECC_EXPORT int ecc_group_size(const ecc_group* g) { return g ? g->size() : 0; }

ECC_EXPORT int ecc_group_ctx(ecc_group* g, int rank, ecc_ctx** ctx)
{
    if (!g || !ctx || rank < 0 || rank >= g->size()) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "bad group / rank");
    *ctx = g->ctxs[rank];
    return ECC_OK;
}

ECC_EXPORT int ecc_group_radon_compute_batch(ecc_group* g, const float* images, int n, int n_u, int n_v, int n_alpha,
                                             int n_t, int filter, int post_process, ecc_dtr** out)
{
    if (!g || !images || !out || n < 1) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "null argument");
    const int G = g->size();
    const int chunk = (n + G - 1) / G;  // the view split of sharding.view_range
    for (int k = 0; k < n; ++k) out[k] = nullptr;
    int rc = run_all(g, [=](int r) -> int {
        const int lo = std::min(r * chunk, n), hi = std::min(lo + chunk, n);
        if (hi <= lo) return ECC_OK;
        int e = ecc_radon_compute_batch(g->ctxs[r], images + (size_t)lo * n_u * n_v, 0, hi - lo, n_u, n_v, n_alpha, n_t, filter,
                                        post_process, out + lo);
        if (e == ECC_OK) e = ecc_ctx_synchronize(g->ctxs[r]);
        return e;
    });
    if (rc != ECC_OK)
        for (int k = 0; k < n; ++k) {
            if (out[k]) ecc_dtr_destroy(out[k]);
            out[k] = nullptr;
        }
    return rc;
}

// ---- the metric over a group ------------------------------------------------------------------------------------
struct ecc_group_metric {
    ecc_group* g = nullptr;
    int n_dtrs = 0, n_views = 0;
    int n_alpha = 0, n_t = 0, n_u = 0, n_v = 0, filter = 0;
    std::vector<float*> replicas;                // per rank: n_dtrs slabs on that rank's device (null: all dtrs aliased)
    std::vector<std::vector<ecc_dtr*>> dtrs;     // per rank: handles (wrapping the replicas or the caller's slabs)
    std::vector<ecc_metric*> metrics;            // per rank
    std::vector<double> partial;                 // per rank, written by the rank's thread
    std::vector<float> pair_values;              // host, all pairs (cost image only)
    // setProjectionMatrices is deferred to the next evaluation: matrices and shard then reach a rank's thread in ONE
    // hand-off (the optimiser pattern is set, evaluate, set, evaluate, ...)
    std::vector<double> pending_Ps;
    bool pending = false;
    // cost-balanced shard boundaries (ecc_pair_shards_balanced), fixed at the first evaluation / by ..._rebalance
    std::vector<int64_t> bounds;
    int bounds_views = 0;
};

ECC_EXPORT int ecc_group_metric_destroy(ecc_group_metric* gm)
{
    if (!gm) return ECC_OK;
    ecc_group* g = gm->g;
    for (size_t r = 0; r < gm->metrics.size(); ++r)
        if (gm->metrics[r]) ecc_metric_destroy(gm->metrics[r]);
    for (size_t r = 0; r < gm->dtrs.size(); ++r)
        for (ecc_dtr* d : gm->dtrs[r])
            if (d) ecc_dtr_destroy(d);
    for (size_t r = 0; r < gm->replicas.size(); ++r)
        if (gm->replicas[r]) {
            (void)hipSetDevice(g->devices[r]);
            (void)hipFree(gm->replicas[r]);
        }
    (void)hipSetDevice(g->devices[0]);
    delete gm;
    return ECC_OK;
}

ECC_EXPORT int ecc_group_metric_create(ecc_group* g, int n_dtrs, ecc_dtr* const* dtrs, ecc_group_metric** out)
{
    if (!g || !dtrs || !out) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (n_dtrs < 1) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "need at least one Radon intermediate");
    const int G = g->size();
    std::unique_ptr<ecc_group_metric> gm(new (std::nothrow) ecc_group_metric());
    if (!gm) return ecc_set_error(ECC_ERR_OUT_OF_MEMORY, "host allocation failed");
    gm->g = g;
    gm->n_dtrs = n_dtrs;
    // where every source dtr lives
    std::vector<float*> src_base(n_dtrs);
    std::vector<int> src_dev(n_dtrs);
    int pitch = 0, rows = 0;
    for (int k = 0; k < n_dtrs; ++k) {
        if (!dtrs[k]) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "null Radon intermediate in list");
        int na, nt, nu, nv, f, p, rw;
        int rc = ecc_dtr_info(dtrs[k], &na, &nt, &nu, &nv, &f, nullptr, nullptr);
        if (rc == ECC_OK) rc = ecc_dtr_device_view(dtrs[k], &src_base[k], &p, &rw);
        if (rc != ECC_OK) return rc;
        if (k == 0) {
            gm->n_alpha = na; gm->n_t = nt; gm->n_u = nu; gm->n_v = nv; gm->filter = f;
            pitch = p; rows = rw;
        } else if (na != gm->n_alpha || nt != gm->n_t) {
            return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "all Radon intermediates must have the same bin counts");
        }
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, src_base[k]) != hipSuccess) {
            (void)hipGetLastError();
            return ecc_set_error(ECC_ERR_HIP, "could not find the device of a Radon intermediate");
        }
        src_dev[k] = attr.device;
    }
    (void)pitch; (void)rows;
    const int64_t slab = ecc_dtr_slab_floats(gm->n_alpha, gm->n_t);
    // the source streams must have produced the data before anyone copies it
    for (int r = 0; r < G; ++r) {
        int rc = ecc_ctx_synchronize(g->ctxs[r]);
        if (rc != ECC_OK) return rc;
    }
    for (int dev = 0, nd = ecc_device_count(); dev < nd; ++dev) {
        bool used = false;
        for (int k = 0; k < n_dtrs; ++k) used = used || src_dev[k] == dev;
        if (used) {
            (void)hipSetDevice(dev);
            if (hipDeviceSynchronize() != hipSuccess) {
                (void)hipGetLastError();
                return ecc_set_error(ECC_ERR_HIP, "hipDeviceSynchronize failed on a source device");
            }
        }
    }
    gm->replicas.assign(G, nullptr);
    gm->dtrs.assign(G, std::vector<ecc_dtr*>(n_dtrs, nullptr));
    gm->metrics.assign(G, nullptr);
    gm->partial.assign(G, 0.0);
    ecc_group_metric* raw = gm.get();
    const int n_alpha = gm->n_alpha, n_t = gm->n_t, n_u = gm->n_u, n_v = gm->n_v, filter = gm->filter;
    int rc = run_all(g, [=, &src_base, &src_dev](int r) -> int {
        const int dev = g->devices[r];
        if (hipSetDevice(dev) != hipSuccess) return ecc_set_error(ECC_ERR_HIP, "hipSetDevice failed");
        bool all_local = true;
        for (int k = 0; k < n_dtrs; ++k) all_local = all_local && src_dev[k] == dev;
        // ecc_group_debug_force_replica (tests): replicate even what is already here, so that the replica path --
        // allocation, copies, probes, metrics on the copy -- runs on a one-GPU box too
        if (g_force_replica.load()) all_local = false;
        if (!all_local) {
            if (hipMalloc((void**)&raw->replicas[r], sizeof(float) * (size_t)slab * n_dtrs) != hipSuccess) {
                (void)hipGetLastError();
                return ecc_set_error(ECC_ERR_OUT_OF_MEMORY, "could not allocate the replica of the Radon-intermediate stack");
            }
            for (int k = 0; k < n_dtrs; ++k) {
                float* dst = raw->replicas[r] + (size_t)slab * k;
                const hipError_t e = src_dev[k] == dev
                                         ? hipMemcpyAsync(dst, src_base[k], sizeof(float) * (size_t)slab, hipMemcpyDeviceToDevice, g->streams[r])
                                         : hipMemcpyPeerAsync(dst, dev, src_base[k], src_dev[k], sizeof(float) * (size_t)slab, g->streams[r]);
                if (e != hipSuccess) {
                    (void)hipGetLastError();
                    return ecc_set_error(ECC_ERR_HIP, (std::string("replicating a Radon intermediate failed: ") + hipGetErrorString(e)).c_str());
                }
            }
        }
        for (int k = 0; k < n_dtrs; ++k) {
            float* base = all_local ? src_base[k] : raw->replicas[r] + (size_t)slab * k;
            const int e = ecc_dtr_wrap_device(g->ctxs[r], base, n_alpha, n_t, n_u, n_v, filter, &raw->dtrs[r][k]);
            if (e != ECC_OK) return e;
        }
        const int e = ecc_metric_create(g->ctxs[r], n_dtrs, raw->dtrs[r].data(), &raw->metrics[r]);  // synchronises the stream
        if (e != ECC_OK || all_local) return e;
        // The replica must BE the stack: three probes (first, middle, last Radon intermediate, 1 KB from the middle of
        // each) read back from the copy and from its source.  A peer copy that silently did something else (first contact
        // with a multi-GPU box, a fabric without peer access, ...) fails here and not as a wrong metric value later.
        const int probes[3] = {0, n_dtrs / 2, n_dtrs - 1};
        float a[256], b[256];
        const size_t off = (size_t)slab / 2, cnt = slab - (int64_t)off < 256 ? (size_t)(slab - (int64_t)off) : 256;
        for (int q = 0; q < 3; ++q) {
            const int k = probes[q];
            if (hipMemcpy(a, raw->replicas[r] + (size_t)slab * k + off, sizeof(float) * cnt, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(b, src_base[k] + off, sizeof(float) * cnt, hipMemcpyDeviceToHost) != hipSuccess) {
                (void)hipGetLastError();
                return ecc_set_error(ECC_ERR_HIP, "could not read back a probe of the replicated Radon-intermediate stack");
            }
            if (std::memcmp(a, b, sizeof(float) * cnt) != 0)
                return ecc_set_error(ECC_ERR_HIP, "the replica of the Radon-intermediate stack differs from its source (device-to-device copy)");
        }
        return ECC_OK;
    });
    if (rc != ECC_OK) {
        ecc_group_metric_destroy(gm.release());
        return rc;
    }
    *out = gm.release();
    return ECC_OK;
}

ECC_EXPORT int ecc_group_metric_set_params(ecc_group_metric* gm, double object_radius_mm, double dkappa, int use_corr)
{
    if (!gm) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "group metric is null");
    for (ecc_metric* m : gm->metrics) {
        const int rc = ecc_metric_set_params(m, object_radius_mm, dkappa, use_corr);
        if (rc != ECC_OK) return rc;
    }
    return ECC_OK;
}

ECC_EXPORT int ecc_group_metric_set_sampling(ecc_group_metric* gm, int mode)
{
    if (!gm) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "group metric is null");
    for (ecc_metric* m : gm->metrics) {
        const int rc = ecc_metric_set_sampling(m, mode);
        if (rc != ECC_OK) return rc;
    }
    return ECC_OK;
}

ECC_EXPORT int ecc_group_metric_set_incremental(ecc_group_metric* gm, int enable)
{
    if (!gm) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "group metric is null");
    for (ecc_metric* m : gm->metrics) {
        const int rc = ecc_metric_set_incremental(m, enable);
        if (rc != ECC_OK) return rc;
    }
    return ECC_OK;
}

namespace {
int flush_projections(ecc_group_metric* gm)
{
    if (!gm->pending) return ECC_OK;
    ecc_group* g = gm->g;
    const double* Ps = gm->pending_Ps.data();
    const int n_views = gm->n_views;
    const int rc = run_all(g, [=](int r) -> int { return ecc_metric_set_projections(gm->metrics[r], Ps, n_views); });
    if (rc == ECC_OK) gm->pending = false;
    return rc;
}
}  // namespace

ECC_EXPORT int ecc_group_metric_rank_metric(ecc_group_metric* gm, int rank, ecc_metric** m)
{
    if (!gm || !m || rank < 0 || rank >= gm->g->size()) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "bad group metric / rank");
    const int rc = flush_projections(gm);  // the rank's metric is about to be used directly: hand over pending matrices
    if (rc != ECC_OK) return rc;
    *m = gm->metrics[rank];
    return ECC_OK;
}

ECC_EXPORT int ecc_group_metric_get_object_radius(ecc_group_metric* gm, double* radius_mm)
{
    if (!gm || !radius_mm) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "null argument");
    const int rc = flush_projections(gm);
    if (rc != ECC_OK) return rc;
    return ecc_metric_get_object_radius(gm->metrics[0], radius_mm);
}

namespace {

// One sharded evaluation: every rank hands the (new) matrices to its device and evaluates its shard; the partial
// sums are added in rank order.  Ps == null: keep the matrices of the last call.
int group_evaluate(ecc_group_metric* gm, const double* Ps, int n_views, float* cost_nxn, double* mean)
{
    ecc_group* g = gm->g;
    const int G = g->size();
    const int64_t n = n_views, n_pairs = n * (n - 1) / 2;
    if (n < 2) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "need at least two views (the reference divides 0/0 here)");
    float* vals = nullptr;
    if (cost_nxn) {
        gm->pair_values.resize((size_t)n_pairs);
        vals = gm->pair_values.data();
    }
    if (gm->bounds_views != n_views || (int)gm->bounds.size() != G + 1) {
        // first evaluation with this many views: fix the shard boundaries from the matrices at hand
        const double* P = Ps ? Ps : nullptr;
        std::vector<int64_t> b((size_t)G + 1);
        int e = ECC_OK;
        if (P) {
            double radius = 0;
            e = ecc_metric_set_projections(gm->metrics[0], P, n_views);  // rank 0 knows the matrices for the radius estimate
            if (e == ECC_OK) e = ecc_metric_get_object_radius(gm->metrics[0], &radius);
            if (e == ECC_OK) e = ecc_pair_shards_balanced(P, n_views, radius, G, b.data());
        } else {
            e = ecc_metric_balanced_shards(gm->metrics[0], G, b.data());
        }
        if (e != ECC_OK) return e;
        gm->bounds = b;
        gm->bounds_views = n_views;
    }
    const int64_t* bounds = gm->bounds.data();
    int rc = run_all(g, [=](int r) -> int {
        int e = ECC_OK;
        if (Ps) e = ecc_metric_set_projections(gm->metrics[r], Ps, n_views);
        if (e != ECC_OK) return e;
        const int64_t first = bounds[r], count = bounds[r + 1] - bounds[r];
        return ecc_metric_evaluate_range(gm->metrics[r], first, count, vals ? vals + first : nullptr, &gm->partial[r]);
    });
    if (rc != ECC_OK) return rc;
    double sum = 0.0;
    for (int r = 0; r < G; ++r) sum += gm->partial[r];  // rank order: the same bits every time
    if (cost_nxn) {
        // entry (i, j), i < j, at index i + j*n; everything else untouched (ref: ...RadonIntermediate.cpp:183,214-221)
        int64_t q = 0;
        for (int64_t i = 0; i < n; ++i)
            for (int64_t j = i + 1; j < n; ++j) cost_nxn[i + j * n] = vals[q++];
    }
    if (mean) *mean = sum / (double)n_pairs;
    return ECC_OK;
}

}  // namespace

// Independent evaluations (a sweep of poses as in Gui/Visualization.h:78-98 plotCostFunction, finite-difference
// gradients): pose p goes to rank p mod G, every rank evaluates ALL pairs of its poses on its own device -- nothing is
// exchanged at all, the throughput is G times one device's ("shard the sweep points").
ECC_EXPORT int ecc_group_metric_evaluate_poses(ecc_group_metric* gm, int n_poses, const double* Ps_batch, int n_views,
                                               double* means)
{
    if (!gm || !Ps_batch || !means) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (n_poses < 1 || n_views < 2) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "need one pose of two views at least");
    ecc_group* g = gm->g;
    const int G = g->size();
    // rank r: the poses r, r + G, ... on its own metric -- batched where they are small deltas of one another (ecc_poses.hip)
    const int rc = run_all(g, [=](int r) -> int {
        return ecc_metric_evaluate_poses_strided(gm->metrics[r], n_poses, Ps_batch, n_views, r, G, means);
    });
    // the ranks' metrics now hold the poses' matrices: the matrices of the last ecc_group_metric_set_projections stay the
    // group's current ones (as the header says) and are handed over again by the next sharded evaluation
    if (!gm->pending_Ps.empty()) gm->pending = true;
    return rc;
}

ECC_EXPORT int ecc_group_metric_rebalance(ecc_group_metric* gm)
{
    if (!gm) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "group metric is null");
    gm->bounds.clear();  // recomputed by the next evaluation from the matrices it runs with
    gm->bounds_views = 0;
    return ECC_OK;
}

ECC_EXPORT int ecc_group_metric_set_projections(ecc_group_metric* gm, const double* Ps, int n_views)
{
    if (!gm || !Ps) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (n_views < 1) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "need at least one projection matrix");
    gm->pending_Ps.assign(Ps, Ps + 12 * (size_t)n_views);
    gm->n_views = n_views;
    gm->pending = true;
    return ECC_OK;
}

ECC_EXPORT int ecc_group_metric_evaluate_all(ecc_group_metric* gm, float* cost_nxn, double* mean)
{
    if (!gm || !mean) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (gm->n_views < 1) return ecc_set_error(ECC_ERR_INVALID_ARGUMENT, "projection matrices have not been set");
    const int rc = group_evaluate(gm, gm->pending ? gm->pending_Ps.data() : nullptr, gm->n_views, cost_nxn, mean);
    if (rc == ECC_OK) gm->pending = false;
    return rc;
}
