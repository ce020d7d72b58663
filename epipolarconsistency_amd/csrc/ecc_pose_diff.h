// ecc_pose_diff.h -- which views of a pose differ from a base set of matrices (host only, no HIP: compiled by
// tests/c/tsan_pose_diff.cpp under ThreadSanitizer as well as by ecc_poses.hip).
//
// ecc_metric_evaluate_poses[_strided] takes FULL matrices per pose (ref for the caller's pattern: Gui/Visualization.h:59-112 --
// the sweep replaces ONE view's matrix per step) and finds the moved views itself.  600 poses of 400 views are 23 MB to compare:
// one thread reads them in ~2 ms, which is what the whole batch takes on the device -- so the comparison runs on up to `max_threads`
// threads, a contiguous run of poses each; the merged result does not depend on the number of threads.
#ifndef ECC_POSE_DIFF_H
#define ECC_POSE_DIFF_H

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace ecc_pose_diff {

struct Result {
    std::vector<int32_t> off;     // per small-delta pose: first entry of `views` (off[0] = 0, one more entry at the end)
    std::vector<int32_t> views;   // the moved views, ascending within a pose
    std::vector<int> batch_pose;  // which pose (an element of `mine`) each entry of off belongs to, in the order of `mine`
    std::vector<int> rest;        // poses with more than max_moved differing views, in the order of `mine`
};

// Ps_batch: poses of n_views x 12 doubles each; mine: the poses to look at; base: n_views x 12 doubles.
// A pose with at most max_moved views whose 12 doubles differ BITWISE from the base's is a small delta.
inline void diff(const double* Ps_batch, int n_views, const std::vector<int>& mine, const double* base, int max_moved,
                 unsigned max_threads, Result* out)
{
    const size_t pose_doubles = 12 * (size_t)n_views, count = mine.size();
    const size_t bytes = count * pose_doubles * sizeof(double);
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t T = (bytes < ((size_t)4 << 20) || max_threads < 2) ? 1 : std::min<size_t>({(size_t)max_threads, (size_t)hw, count / 32 + 1});
    std::vector<std::vector<int32_t>> part_views(T);
    std::vector<std::vector<int32_t>> part_count(T);  // per pose of the part: moved views, or -1 = not a small delta
    auto work = [&](size_t t) {
        const size_t lo = count * t / T, hi = count * (t + 1) / T;
        std::vector<int32_t>&pv = part_views[t], &pc = part_count[t];
        pc.reserve(hi - lo);
        for (size_t q = lo; q < hi; ++q) {
            const double* Pp = Ps_batch + pose_doubles * (size_t)mine[q];
            const size_t before = pv.size();
            // blocks of eight views first: a sweep's poses differ from the base in one view or two
            for (int v0 = 0; v0 < n_views && pv.size() - before <= (size_t)max_moved; v0 += 8) {
                const int v1 = std::min(n_views, v0 + 8);
                if (std::memcmp(Pp + 12 * (size_t)v0, base + 12 * (size_t)v0, sizeof(double) * 12 * (size_t)(v1 - v0)) == 0) continue;
                for (int v = v0; v < v1; ++v)
                    if (std::memcmp(Pp + 12 * (size_t)v, base + 12 * (size_t)v, sizeof(double) * 12) != 0) pv.push_back(v);
            }
            if (pv.size() - before > (size_t)max_moved) {
                pv.resize(before);
                pc.push_back(-1);
            } else pc.push_back((int32_t)(pv.size() - before));
        }
    };
    if (T == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (size_t t = 1; t < T; ++t) th.emplace_back(work, t);
        work(0);
        for (std::thread& x : th) x.join();
    }
    out->off.assign(1, 0);
    out->views.clear();
    out->batch_pose.clear();
    out->rest.clear();
    for (size_t t = 0; t < T; ++t) {
        const size_t lo = count * t / T;
        size_t at = 0;
        for (size_t q = 0; q < part_count[t].size(); ++q) {
            const int32_t c = part_count[t][q];
            if (c < 0) out->rest.push_back(mine[lo + q]);
            else {
                out->views.insert(out->views.end(), part_views[t].begin() + at, part_views[t].begin() + at + c);
                at += (size_t)c;
                out->batch_pose.push_back(mine[lo + q]);
                out->off.push_back((int32_t)out->views.size());
            }
        }
    }
}

}  // namespace ecc_pose_diff

#endif
