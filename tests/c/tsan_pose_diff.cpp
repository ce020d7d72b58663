// ThreadSanitizer driver for the host-side comparison of the batched pose evaluation (epipolarconsistency_amd/csrc/ecc_pose_diff.h),
// CPU only: 300 poses of 200 views (5.8 MB: the multi-threaded form) with 0, 1, 2, 33 and all views moved, strided pose lists,
// eight threads against one -- the merged result must not depend on the number of threads.  Built and run by scripts/sanitize.sh
// with -fsanitize=thread; also a plain correctness check (tests/test_abi_and_host.py builds it without the sanitizer).
#include <cstdio>
#include <cstdlib>

#include "../../epipolarconsistency_amd/csrc/ecc_pose_diff.h"

int main()
{
    const int n = 200, K = 300, max_moved = 32;
    std::vector<double> base(12 * (size_t)n), batch(12 * (size_t)n * K);
    unsigned long long state = 88172645463325252ull;
    auto rnd = [&]() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; };
    for (double& x : base) x = (double)(rnd() % 100000) / 7.0;
    std::vector<std::vector<int>> truth(K);
    for (int k = 0; k < K; ++k) {
        double* P = batch.data() + 12 * (size_t)n * k;
        std::copy(base.begin(), base.end(), P);
        const int kind = k % 6;
        std::vector<int> moved;
        if (kind == 1) moved = {(int)(rnd() % n)};
        else if (kind == 2) moved = {3, n - 1};
        else if (kind == 3) for (int v = 0; v < 33; ++v) moved.push_back(2 * v);          // one more than the batch takes
        else if (kind == 4) for (int v = 0; v < n; ++v) moved.push_back(v);               // a different trajectory
        else if (kind == 5) moved = {0, 7, 8, 9, 15, 16, 199};                            // across the blocks of eight
        for (int v : moved) P[12 * (size_t)v + (rnd() % 12)] += 1.0;
        std::sort(moved.begin(), moved.end());
        truth[k] = moved;
    }
    int bad = 0;
    for (int stride = 1; stride <= 3; ++stride) {
        std::vector<int> mine;
        for (int p = stride - 1; p < K; p += stride) mine.push_back(p);
        ecc_pose_diff::Result one, many;
        ecc_pose_diff::diff(batch.data(), n, mine, base.data(), max_moved, 1, &one);
        ecc_pose_diff::diff(batch.data(), n, mine, base.data(), max_moved, 8, &many);
        if (one.off != many.off || one.views != many.views || one.batch_pose != many.batch_pose || one.rest != many.rest) ++bad;
        size_t b = 0, r = 0;
        for (int p : mine) {
            if ((int)truth[p].size() > max_moved) {
                if (r >= many.rest.size() || many.rest[r++] != p) ++bad;
            } else {
                if (b >= many.batch_pose.size() || many.batch_pose[b] != p) { ++bad; continue; }
                const std::vector<int32_t> got(many.views.begin() + many.off[b], many.views.begin() + many.off[b + 1]);
                if (got != std::vector<int32_t>(truth[p].begin(), truth[p].end())) ++bad;
                ++b;
            }
        }
        if (b != many.batch_pose.size() || r != many.rest.size()) ++bad;
    }
    std::printf("tsan_pose_diff: %d poses of %d views, strides 1-3, 1 and 8 threads: %s\n", K, n, bad ? "MISMATCH" : "ok");
    return bad ? 1 : 0;
}
