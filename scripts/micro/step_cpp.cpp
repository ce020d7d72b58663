// One optimiser step from a C++ caller (what the reference's callers are: Gui/SingleImageMotion.h:84-90 -> setProjectionMatrices +
// evaluate), through the C ABI: how much of bench.py's step is the Python caller.  Random Radon intermediates (the pair
// kernel's time depends on the geometry, not on the values), the benchmark's circular short scan, one view moved per step.
//   g++ -O2 -std=c++11 -Iinclude scripts/micro/step_cpp.cpp -Lepipolarconsistency_amd -lecc_hip -Wl,-rpath,$PWD/epipolarconsistency_amd -o /tmp/step_cpp
//   /tmp/step_cpp [views=400] [size=1024] [bins=768] [steps=400]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ecc_hip.h"

#define CHECK(expr)                                                              \
    do {                                                                         \
        if ((expr) != ECC_OK) {                                                  \
            std::fprintf(stderr, "%s: %s\n", #expr, ecc_last_error());           \
            return 1;                                                            \
        }                                                                        \
    } while (0)

static void make_P(double phi, int S, double pixel_mm, double* P)  // column-major 3x4, source on a circle around the y axis
{
    const double sid = 744.3, sdd = 1088.15, f = sdd / pixel_mm;
    const double c = std::cos(phi), s = std::sin(phi);
    const double R[3][4] = {{-s, 0, c, 0}, {0, 1, 0, 0}, {-c, 0, -s, sid}};
    const double K[3][3] = {{f, 0, S * 0.5}, {0, f, S * 0.5}, {0, 0, 1}};
    for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 4; ++q) {
            double v = 0;
            for (int k = 0; k < 3; ++k) v += K[r][k] * R[k][q];
            P[q * 3 + r] = v;
        }
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 400, S = argc > 2 ? std::atoi(argv[2]) : 1024, B = argc > 3 ? std::atoi(argv[3]) : 768;
    const int steps = argc > 4 ? std::atoi(argv[4]) : 400;
    ecc_ctx* ctx = nullptr;
    CHECK(ecc_ctx_create(0, nullptr, &ctx));
    std::vector<float> data((size_t)B * B);
    unsigned seed = 12345u;
    std::vector<ecc_dtr*> base(8, nullptr), dtrs(n, nullptr);
    for (size_t k = 0; k < base.size(); ++k) {
        for (float& v : data) { seed = seed * 1664525u + 1013904223u; v = (float)((seed >> 8) & 0xffff) / 65536.f - 0.5f; }
        CHECK(ecc_dtr_from_host(ctx, data.data(), B, B, S, S, ECC_FILTER_DERIVATIVE, &base[k]));
    }
    for (int v = 0; v < n; ++v) dtrs[v] = base[v % base.size()];
    ecc_metric* m = nullptr;
    CHECK(ecc_metric_create(ctx, n, dtrs.data(), &m));
    const double pixel_mm = 0.308 * 1024.0 / S, max_angle = 199.7 * 3.14159265358979323846 / 180.0;  // (not 200: views exactly 180 degrees apart have their baseline through the origin -- 0/0 in computeK01, in the reference too)
    std::vector<double> Ps((size_t)12 * n), pose((size_t)12 * n);
    for (int v = 0; v < n; ++v) make_P(max_angle * v / n, S, pixel_mm, &Ps[(size_t)12 * v]);
    double mean = 0, first = 0;
    for (int pass = 0; pass < 2; ++pass) {  // pass 0: warm-up
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < steps; ++k) {
            pose = Ps;
            pose[(size_t)12 * (n / 2) + 9] += 0.01 * (k % 50);  // the translation column of view n/2
            CHECK(ecc_metric_set_projections(m, pose.data(), n));
            CHECK(ecc_metric_evaluate_all(m, nullptr, &mean));
            if (k == 0) first = mean;
        }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (pass) std::printf("C++ caller: %d views %dx%d, %d pairs: %.1f us per step, %.0f evaluations/s (mean %.9g ... %.9g)\n", n, S, S,
                              n * (n - 1) / 2, 1e6 * s / steps, steps / s, first, mean);
    }
    ecc_metric_destroy(m);
    for (ecc_dtr* d : base) ecc_dtr_destroy(d);
    ecc_ctx_destroy(ctx);
    return 0;
}
