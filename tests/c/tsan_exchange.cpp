// ThreadSanitizer driver for the shared-memory scalar exchange (epipolarconsistency_amd/csrc/ecc_exchange.cpp), CPU only:
// the ranks of one job as threads of one process (the sanitizer does not see across processes), 2000 generations of
// ecc_exchange_sum, every rank checking every total.  Built and run by scripts/sanitize.sh with -fsanitize=thread.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/ecc_hip.h"

static thread_local std::string g_err;
extern "C" int ecc_set_error(int code, const char* msg)  // the library's error slot (ecc_capi.hip) stands outside this build
{
    g_err = msg ? msg : "";
    return code;
}

int main()
{
    const int world = 4, generations = 2000;
    const std::string name = "/ecc_tsan_" + std::to_string((long long)getpid());
    std::vector<ecc_exchange*> ex(world, nullptr);
    if (ecc_exchange_open(name.c_str(), 0, world, &ex[0]) != ECC_OK) {
        std::printf("tsan_exchange: cannot open the exchange (%s)\n", g_err.c_str());
        return 2;
    }
    std::atomic<int> failures{0};
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r)
        th.emplace_back([&, r] {
            if (r != 0 && ecc_exchange_open(name.c_str(), r, world, &ex[r]) != ECC_OK) {
                failures.fetch_add(1);
                return;
            }
            for (int g = 0; g < generations; ++g) {
                double total = 0;
                if (ecc_exchange_sum(ex[r], (double)(r + 1) * (g + 1), &total) != ECC_OK || total != 10.0 * (g + 1)) {
                    failures.fetch_add(1);
                    return;
                }
            }
        });
    for (std::thread& t : th) t.join();
    for (int r = 0; r < world; ++r)
        if (ex[r]) ecc_exchange_close(ex[r]);
    std::printf("tsan_exchange: %d failures\n", failures.load());
    return failures.load() ? 1 : 0;
}
