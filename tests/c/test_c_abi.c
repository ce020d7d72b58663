/* The C ABI from plain C (gcc -std=c99 -pedantic): include/ecc_hip.h has to be a C header, and a C program linking
 * libecc_hip.so can run the whole path: images -> Radon intermediates -> all-pairs metric, pose-delta mode included.
 * Without a device (argument "nodevice") only the host-side entry points are exercised. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ecc_hip.h"

#define CHECK(call)                                                             \
    do {                                                                        \
        int rc_ = (call);                                                       \
        if (rc_ != ECC_OK) {                                                    \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ecc_last_error());    \
            return 1;                                                           \
        }                                                                       \
    } while (0)

/* column-major 3x4 P = K [R | t] of a view on a circle around the y axis looking at the origin */
static void make_P(double phi, int n_u, int n_v, double* P)
{
    const double f = 900.0, sid = 700.0, c = cos(phi), s = sin(phi);
    const double R[9] = {c, 0, -s, 0, 1, 0, s, 0, c}; /* rows */
    const double t[3] = {0, 0, sid};
    const double K[9] = {f, 0, 0.5 * n_u, 0, f, 0.5 * n_v, 0, 0, 1};
    int r, cc, k;
    for (r = 0; r < 3; ++r)
        for (cc = 0; cc < 4; ++cc) {
            double v = 0;
            for (k = 0; k < 3; ++k) v += K[3 * r + k] * (cc < 3 ? R[3 * k + cc] : t[k]);
            P[r + 3 * cc] = v;
        }
}

int main(int argc, char** argv)
{
    enum { N = 6, NU = 96, NV = 80, NA = 64, NT = 64 };
    static float imgs[N * NU * NV];
    double Ps[12 * N], radius, k0, k1, mean_full, mean_inc, mean_moved_full, mean_moved_inc;
    int i, j, v, x, y;
    int64_t recomputed = 0, first = 0, count = 0;
    ecc_ctx* ctx = NULL;
    ecc_dtr* dtrs[N];
    ecc_metric* m = NULL;
    printf("version %d devices %d\n", ecc_version(), ecc_device_count());
    for (v = 0; v < N; ++v) make_P(0.55 * v, NU, NV, Ps + 12 * v);
    radius = ecc_host_object_radius(Ps, NU, NV);
    ecc_host_angular_range(Ps, Ps + 12, radius, &k0, &k1);
    ecc_get_ij(7, N, &i, &j);
    ecc_pair_shard(15, 4, 3, &first, &count);
    printf("host radius %.6f range %.6f %.6f ij %d %d shard %lld %lld slab %lld\n", radius, k0, k1, i, j, (long long)first,
           (long long)count, (long long)ecc_dtr_slab_floats(NA, NT));
    if (!(radius > 0 && k0 < 0 && k1 > 0 && i == 1 && j == 4 && first + count == 15)) return 1;
    if (argc > 1 && strcmp(argv[1], "nodevice") == 0) {
        if (ecc_ctx_create(0, NULL, &ctx) == ECC_OK && ecc_device_count() == 0) return 1; /* must fail loudly without a GPU */
        return 0;
    }
    for (v = 0; v < N; ++v) /* a blob that moves with the view: any non-trivial image will do */
        for (y = 0; y < NV; ++y)
            for (x = 0; x < NU; ++x) {
                const double dx = x - 48.0 - 10.0 * cos(0.55 * v), dy = y - 40.0 - 4.0 * sin(0.9 * v);
                imgs[(v * NV + y) * NU + x] = (float)exp(-(dx * dx + dy * dy) / 200.0);
            }
    CHECK(ecc_ctx_create(0, NULL, &ctx));
    CHECK(ecc_radon_compute_batch(ctx, imgs, 0, N, NU, NV, NA, NT, ECC_FILTER_DERIVATIVE, ECC_POST_IDENTITY, dtrs));
    CHECK(ecc_metric_create(ctx, N, dtrs, &m));
    CHECK(ecc_metric_set_sampling(m, ECC_SAMPLING_POLYNOMIAL));
    CHECK(ecc_metric_set_projections(m, Ps, N));
    CHECK(ecc_metric_evaluate_all(m, NULL, &mean_full));
    CHECK(ecc_metric_set_incremental(m, 1));
    CHECK(ecc_metric_evaluate_all(m, NULL, &mean_inc));
    make_P(0.55 * 3 + 0.01, NU, NV, Ps + 12 * 3); /* one view moves */
    CHECK(ecc_metric_set_projections(m, Ps, N));
    CHECK(ecc_metric_evaluate_all(m, NULL, &mean_moved_inc));
    CHECK(ecc_metric_last_evaluated_pairs(m, &recomputed));
    CHECK(ecc_metric_set_incremental(m, 0));
    CHECK(ecc_metric_evaluate_all(m, NULL, &mean_moved_full));
    printf("mean %.17g moved %.17g recomputed %lld\n", mean_full, mean_moved_full, (long long)recomputed);
    if (!(mean_full > 0) || mean_inc != mean_full || mean_moved_inc != mean_moved_full || mean_moved_full == mean_full ||
        recomputed != N - 1)
        return 1;
    {   /* many poses per call (csrc/ecc_poses.hip; ref for the pattern: Gui/Visualization.h:59-112): three poses as deltas of the
           current matrices -- view 3 back at its first angle, view 1 moved, both -- against one ecc_metric_set_projections +
           ecc_metric_evaluate_all each: the same bits; then the same three as full matrix sets */
        int32_t off[4] = {0, 1, 2, 4}, views[4] = {3, 1, 1, 3};
        double moved[4 * 12], batch[3 * N * 12], means[3], dense[3], one[3], keep[N * 12];
        int64_t batched = 0;
        int k;
        memcpy(keep, Ps, sizeof(keep));
        make_P(0.55 * 3, NU, NV, moved + 0);
        make_P(0.55 * 1 - 0.02, NU, NV, moved + 12);
        memcpy(moved + 24, moved + 12, sizeof(double) * 12);
        memcpy(moved + 36, moved + 0, sizeof(double) * 12);
        CHECK(ecc_metric_evaluate_pose_deltas(m, 3, off, views, moved, means));
        CHECK(ecc_metric_last_batched_poses(m, &batched));
        for (k = 0; k < 3; ++k) {
            int q;
            memcpy(batch + k * N * 12, keep, sizeof(keep));
            for (q = off[k]; q < off[k + 1]; ++q) memcpy(batch + k * N * 12 + 12 * views[q], moved + 12 * q, sizeof(double) * 12);
        }
        CHECK(ecc_metric_evaluate_poses(m, 3, batch, N, dense));
        CHECK(ecc_metric_set_pose_batching(m, 0));
        for (k = 0; k < 3; ++k) {
            CHECK(ecc_metric_set_projections(m, batch + k * N * 12, N));
            CHECK(ecc_metric_evaluate_all(m, NULL, &one[k]));
        }
        printf("poses %.17g %.17g %.17g batched %lld\n", means[0], means[1], means[2], (long long)batched);
        for (k = 0; k < 3; ++k)
            if (means[k] != one[k] || dense[k] != one[k]) return 1;
        if (batched != 3 || means[0] != mean_full || means[1] == means[0]) return 1;
    }
    CHECK(ecc_metric_destroy(m));
    for (v = 0; v < N; ++v) CHECK(ecc_dtr_destroy(dtrs[v]));
    CHECK(ecc_ctx_destroy(ctx));
    printf("ok\n");
    return 0;
}
