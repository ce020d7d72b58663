// ThreadSanitizer driver for the group's job hand-off (epipolarconsistency_amd/csrc/ecc_worker_pool.h), CPU only:
// 8 ranks, thousands of back-to-back jobs (workers spinning), jobs separated by pauses (workers asleep on the condition
// variable), failing and throwing ranks.  Built and run by scripts/sanitize.sh with -fsanitize=thread.
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

#include "../../epipolarconsistency_amd/csrc/ecc_worker_pool.h"

int main()
{
    const int n = 8;
    EccWorkerPool pool;
    std::vector<int> inits(n, 0);
    pool.start(n, [&](int r) { inits[r] = 1; }, [] { return std::string("status message"); });
    std::vector<long long> per_rank(n, 0);  // written by rank r only, read by the caller after run_all
    long long shared_plain = 0;              // written by the caller between jobs, read by every rank inside a job
    std::atomic<long long> total{0};
    int failures = 0;
    for (int it = 0; it < 20000; ++it) {
        shared_plain = it;
        const int bad = pool.run_all([&](int r) -> int {
            per_rank[r] += shared_plain + r;
            total.fetch_add(1, std::memory_order_relaxed);
            return 0;
        }, -7);
        if (bad != -1) ++failures;
        if (it % 4000 == 3999) std::this_thread::sleep_for(std::chrono::milliseconds(3));  // the workers go to sleep
    }
    for (int r = 0; r < n; ++r) {
        long long want = 0;
        for (int it = 0; it < 20000; ++it) want += it + r;
        if (per_rank[r] != want) ++failures;
    }
    if (total.load() != 20000LL * n) ++failures;
    // a failing rank and a throwing rank
    int bad = pool.run_all([&](int r) -> int { return r == 5 ? 42 : 0; }, -7);
    if (bad != 5 || pool.status(5) != 42 || pool.message(5) != "status message") ++failures;
    bad = pool.run_all([&](int r) -> int {
        if (r == 3) throw std::runtime_error("boom");
        return 0;
    }, -7);
    if (bad != 3 || pool.status(3) != -7 || pool.message(3) != "boom") ++failures;
    bad = pool.run_all([&](int) -> int { return 0; }, -7);
    if (bad != -1) ++failures;
    pool.stop();
    for (int r = 1; r < n; ++r)
        if (!inits[r]) ++failures;
    std::printf("tsan_worker_pool: %d failures\n", failures);
    return failures ? 1 : 0;
}
