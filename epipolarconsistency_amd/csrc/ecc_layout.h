// ecc_layout.h -- private HBM layout of one Radon intermediate ("dtr") and shared kernel params.
//
// The reference keeps a dtr as a CUDA texture over an n_t x n_alpha, alpha-fast array
// (ref: LibEpipolarConsistency/RadonIntermediate.cu:44, RadonIntermediate.cpp:188-196) and lets the
// texture unit do clamp addressing + bilinear filtering.  Here a dtr is a plain float slab laid
// out for the pair kernel's gather pattern:
//
//   element (ix = angle bin, iy = distance bin)  ->  base[(ix + 1) * pitch + (iy + 1)]
//
//   * t (distance) is the FAST axis: along one epipolar-plane sweep the sampled line moves mostly
//     in distance, so the 64 lanes of a wave (consecutive kappa samples) read a handful of
//     contiguous 128-B lines instead of 64 different ones;
//   * one replicated border row/column on every side (rows -1, n_alpha; columns -1, n_t) turns
//     the texture unit's clamp addressing into plain in-bounds loads: a bilinear footprint never
//     needs per-tap index clamps;
//   * pitch is a multiple of 32 floats so every angle row starts on a 128-B line.
#ifndef ECC_LAYOUT_H
#define ECC_LAYOUT_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
// ref: EpipolarConsistencyCommon.hxx:52-79 (get_ij), closed form: pairs before row i = i*n - i(i+1)/2.
__device__ __forceinline__ void ecc_get_ij_device(long long ij, int n, int& i, int& j)
{
    double nn = (double)n - 0.5;
    int r = (int)floor(nn - sqrt(nn * nn - 2.0 * (double)ij));
    r = max(0, min(r, n - 2));
    // fix-up against rounding of the square root
    while (r > 0 && (long long)r * n - (long long)r * (r + 1) / 2 > ij) --r;
    while ((long long)(r + 1) * n - (long long)(r + 1) * (r + 2) / 2 <= ij) ++r;
    i = r;
    j = (int)(ij - ((long long)r * n - (long long)r * (r + 1) / 2)) + r + 1;
}
#endif

static inline int ecc_layout_pitch(int n_t) { return ((n_t + 2) + 31) / 32 * 32; }
static inline int ecc_layout_rows(int n_alpha) { return n_alpha + 2; }
static inline int64_t ecc_layout_floats(int n_alpha, int n_t)
{
    return (int64_t)ecc_layout_rows(n_alpha) * ecc_layout_pitch(n_t);
}

// ---- Radon-intermediate kernel -------------------------------------------------------------
struct EccRadonParams {
    const float* images;   // n_img * n_v * n_u floats (row-major, x fastest)
    const float* imagesT;  // the same images transposed (n_u rows of n_v floats each)
    float* out;            // n_img slabs in the private layout
    const float* trig;     // 2 * n_alpha floats: (sinf(alpha), cosf(alpha)), alpha = (ix/n_alpha - .5)*Pi
    int64_t image_stride;  // floats between images
    int64_t out_stride;    // floats between slabs
    int n_img;
    int n_u, n_v;
    int n_alpha, n_t;
    int pitch;
    int post_process;
    int arithmetic;        // 0: exact (unfused) sampling loop, 1: contracted (ECC_RADON_FMA)
};

// ---- pair kernel ---------------------------------------------------------------------------
// Degree of the per-pair, per-view polynomials xa(kappa), yd(kappa) on [-kappa_max, kappa_max] (see pairs_kernel.hip):
// 11 coefficients each plus a low part of the constant term (the constant is ~n/2 bins; its float rounding alone
// would shift a whole curve by up to 1.5e-5 bins).
#define ECC_SKIP_WORDS 16
#ifndef ECC_PAIRS_SPLIT_MAX
#define ECC_PAIRS_SPLIT_MAX 4096  // launches up to here: several waves per pair (pairs_split_kernel)
#endif
#ifndef ECC_PAIRS_SPLIT4_MAX
#define ECC_PAIRS_SPLIT4_MAX 2048  // ... four of them up to here, two beyond (bench.py --views 64 --size 512, 2016 pairs: 34.5 -> 32.6 us per
                                   // step against two; 2775 and 4095 pairs: the same either way)
#endif
#ifndef ECC_PAIRS_SPLIT8_MAX
#define ECC_PAIRS_SPLIT8_MAX 768   // ... eight (one pair per 512-thread workgroup) up to here
#endif
#ifndef ECC_REFIT_FIRST_MAX_PAIRS
#define ECC_REFIT_FIRST_MAX_PAIRS 16384  // all-pairs launches below this (under two rounds of resident workgroups): the refit's launch goes out first
#endif
#ifndef ECC_BESIDE_ONE_WAVE_MIN_PAIRS
#define ECC_BESIDE_ONE_WAVE_MIN_PAIRS 32768  // all-pairs launches from here on: the moved view's pairs beside them with one wave per pair
#endif
#define ECC_POLY_DEG 10
#define ECC_POLY_CHECKS 3
// The polynomials cover |kappa| <= ecc_kappa_fit(kappa_max); a pair whose range goes on beyond that (kappa_max = pi/2: the baseline
// passes through the object, 3.5 % of the benchmark's pairs) takes its inner samples from the polynomials and the rest from the
// exact per-sample loop.  Over the whole pi/2 a curve of every such pair switches its fold state and the fit is refused; on the
// inner 0.98 rad (62 % of the samples) none does and the fit is good to 3e-7 bins in the median, 1.3e-6 at worst over 300 such
// pairs of the benchmark's scan
// (scripts/analysis/heavy_pairs_inner_range.py).  k01_kernel and the pair kernels both derive the bound from the record's kappa_max.
#ifndef ECC_POLY_KAPPA_FIT_MAX
#define ECC_POLY_KAPPA_FIT_MAX 0.98f
#endif
#if defined(__HIPCC__)
__host__ __device__
#endif
inline float ecc_kappa_fit(float kappa_max) { return kappa_max < ECC_POLY_KAPPA_FIT_MAX ? kappa_max : ECC_POLY_KAPPA_FIT_MAX; }

// What k01_kernel hands to pairs_kernel for one pair (296 bytes, read with scalar loads).
struct EccPairRecord {
    float K0[8];   // ref: computeK01 (EpipolarConsistencyCommon.hxx:93-149): K0[6] baseline distance, K0[7] view angle
    float K1[8];   // K1[6] dkappa, K1[7] kappa_max
    int iD0, iD1;  // Radon intermediates of the two views
    int ci, cj;    // cost-image position (all-pairs mode)
    int poly_ok;   // 0: exact per-sample path; else the sample coordinates of both views are given by the polynomials
                   // below, evaluated up to the degree poly_ok & ~1 (4, 6, 8 or 10; economised in k01_kernel, higher coefficients
                   // are zero); bit 0: k01's bound on the polynomials says no sample can reach a clamp of the pair kernel
    float x_scale; // x = kappa * x_scale in (0, 1], x_scale = 1 / ecc_kappa_fit(kappa_max); the -kappa samples are the same polynomials at -x
    unsigned fold[2];                  // per view: 0x80000000 when the (alpha+pi, -t) fold applies on the +kappa side
                                       // (it is the opposite on the -kappa side: the line is negated there)
    float ca[2][ECC_POLY_DEG + 3];     // angle coordinate (padded texel units): monomial coefficients in x, c0 first;
                                       // [DEG+1] is c0's low part (c0 = c[0] + c[DEG+1]), [DEG+2] the low part to use at
                                       // -x: the reference's float Pi puts the two fold states 2.78e-8 * n_alpha bins apart
    float cd[2][ECC_POLY_DEG + 2];     // distance coordinate, same layout without the last entry
};

// Constant tables of the polynomial fit (float64, device memory).  The Chebyshev nodes are symmetric
// (x[N-1-j] = -x[j], x[H] = 0, N = DEG+1, H = DEG/2), so the fit splits into an even part E(z) of degree H in
// z = x^2 through the H+1 values (f[j] + f[N-1-j])/2 (j <= H) and an odd part x O(z), O of degree H-1 through the H
// values (f[j] - f[N-1-j])/(2 x[j]) (j < H):  e_k = sum_j Ae[k][j] fe[j],  o_k = sum_j Ao[k][j] fo[j].
#define ECC_TRIG_STEPS 32
struct EccPolyTables {
    double nodes[ECC_POLY_DEG + 1];
    double checks[ECC_POLY_CHECKS];
    double Ae[(ECC_POLY_DEG / 2 + 1) * (ECC_POLY_DEG / 2 + 1)];
    double Ao[(ECC_POLY_DEG / 2) * (ECC_POLY_DEG / 2)];
    double sc[ECC_TRIG_STEPS + 1][2];  // {sin, cos}(k pi / (2 ECC_TRIG_STEPS)), k = 0 .. ECC_TRIG_STEPS: the fit's float64 trigonometry
};

struct EccPairParams {
    const float* const* dtrs;  // device table of the dtrs' ROW-PAIRED copies (build_paired_kernel), one per dtr
    const float* Cs;           // 4 floats per view  (source positions, w = 1)
    const float* PinvTs;       // 12 floats per view ((P^+)^T, 3x4 column-major)
    const int32_t* indices;    // optional n_pairs x 4 (P0, P1, dtr0, dtr1); null = all pairs
    float* pair_values;        // optional, `count` floats (local pair order)
    const int32_t* value_slots;  // optional, `count` entries: pair_values[value_slots[k]] instead of pair_values[k]
    int reference_split;       // reference arithmetic: waves per pair, 1 or 4 (4: one pair per workgroup, for few pairs)
    int beside_another_launch; // host only: this launch runs beside a chip-filling one on another stream -- 1: 256-thread workgroups only;
                               // 2: that launch lasts far longer than one pair's wave -- one wave per pair (pairs_kernel), see ecc_launch_pairs
                               // (a 512-thread workgroup waits until eight wave slots of ONE CU are free at once: the moved view's
                               // 399 pairs took 304 us instead of 29 beside the all-pairs launch)
    float* cost;               // optional n x n cost image (index i + j*n)
    float* K01_out;            // optional debug output, 16 floats per pair
    EccPairRecord* records;    // `count` records, written by k01_kernel, read by pairs_kernel
    // Record reuse (ecc_metric_set_record_reuse): k01_kernel runs over an index list of the pairs whose matrices changed
    // and writes each record into its slot of a record array kept from the previous evaluation.
    const int32_t* record_slots;  // optional, `count` entries: records[record_slots[k]] instead of records[k]
    const int32_t* patch_ref;     // optional, 2 per pair: entry of patch_geo that holds view P0 / P1, or -1 (device arrays)
    const float* patch_geo;       // 16 floats per entry: (P^+)^T (12) + C (4) of a view whose matrix changed, computed
                                  // on the host (pinned memory: read over PCIe inside k01_kernel, no copy command)
    const int32_t* patch_views;   // view of each entry; the launch also copies the entries into PinvTs / Cs
    int patch_count;
    // pairs_kernel over all pairs EXCEPT those that contain a view of this set (they are refitted and sampled by a list
    // launch on a second stream meanwhile): bit v of skip_mask = view v changed; views < 32 * ECC_SKIP_WORDS
    int skip_enabled;
    unsigned skip_mask[16];
    const EccPolyTables* poly; // constant tables of the polynomial fit (null: exact path for every pair)
    int64_t first;             // first pair (get_ij order) handled by this launch
    int64_t count;             // pairs handled by this launch
    int n_views;
    int n_alpha, n_t, pitch;
    float n_x2, n_y2;          // half image size
    float object_radius_mm;
    float num_samples;         // n_t*step_t*2.f  (ref: ...RadonIntermediate.cu:320)
    float range_t;             // n_t*step_t      (ref: ...RadonIntermediate.cu:372)
    float dkappa_user;         // <=0: automatic
    int k_limit;               // launch bound on the kappa index (ref: ...RadonIntermediate.cu:348-358)
    int is_derivative;
    int use_corr;              // MetricRadonIntermediate::useCorrelation (ref: ...RadonIntermediate.cu:116-149)
    const float* const* slabs; // device table of the dtrs' slabs (private layout); sampled by ECC_SAMPLING_REFERENCE
    int reference_arithmetic;  // ECC_SAMPLING_REFERENCE: pairs_reference_kernel instead of pairs_kernel
    const float* const* quads; // device table of the dtrs' ROW-QUAD copies (build_quad_kernel) or null
    unsigned quad_group_bytes; // bytes of one group of four rows in a row-quad copy (pitch * 64)
    float economise_tol;       // bins: bound on what lowering a pair's polynomial degree may cost (k01_kernel, economise)
    int wide_offsets;          // a row-paired copy is 2^24 bytes or more: integer instead of fp32 offset arithmetic
};

// ---- one-launch evaluation of small pair sets (small_eval_kernel.hip) ---------------------------
// Where the one launch beats the stream-ordered launches (measured on one MI355X, us per setProjectionMatrices + evaluate,
// Python caller; the stream-ordered path with pairs_split_kernel and pinned index lists, i.e. as it is at the end of round 4):
// all pairs of n views at 512^2 (724 samples per pair): 1 pair 22.7 against 27.3, 28 pairs 25.1 / 28.0, 190 pairs 25.5 / 27.2,
// 528 pairs 30.9 / 25.6; index lists on 400 views at 1024^2 (1448 samples per pair): 1 pair 26.1 / 27.4, 399 pairs 33-40 / 30.0,
// 512 pairs 34.9 / 33.0.  (Every value of the one launch is a PCIe write of its own and every workgroup costs the dispatcher
// 11 ns; before the split kernel and the pinned lists the stream-ordered path took 46 / 54 / 58 us for these lists and the
// one launch won up to 576 pairs.)
#define ECC_SMALL_EVAL_MAX_PAIRS 192
#define ECC_SMALL_EVAL_PAIR_BOUND(k_limit) ECC_SMALL_EVAL_MAX_PAIRS
// ECC_SAMPLING_REFERENCE launches of at most this many pairs use 1024 threads per pair (pairs_reference_wide_kernel, and the
// one-launch form of small_eval_kernel.hip): two such workgroups per CU hold them all at once.  Same bits either way.
#ifndef ECC_REFERENCE_WIDE_MAX_PAIRS
#define ECC_REFERENCE_WIDE_MAX_PAIRS 512
#endif
// launches of at most this many pairs fit their records with ECC_K01_SMALL_LANES lanes per fit (k01_kernel<16>, pairs_kernel.hip:
// the kernel's time is the length of one fit's dependent chain there), and take E1 of a few changed views in their kernel
// arguments instead of an e1_kernel launch (k01_patched_kernel)
#ifndef ECC_K01_WIDE_MAX_PAIRS
#define ECC_K01_WIDE_MAX_PAIRS 4096
#endif
#ifndef ECC_K01_SMALL_LANES
#define ECC_K01_SMALL_LANES 16  // 16: one exact curve point per lane; 8: two or three (round 3)
#endif
// ... up to this many pairs; beyond, 8 lanes per fit (measured: 1 pair / 528 pairs 20.0 / 25.3 -> 18.9 / 23.3 us per evaluation with
// 16 lanes, 1035 and 2016 pairs the same either way, 4095 pairs 42.3 -> 45.2: twice the workgroups for the same chain)
#ifndef ECC_K01_LANES16_MAX_PAIRS
#define ECC_K01_LANES16_MAX_PAIRS 1024
#endif
#define ECC_SMALL_PATCH_MAX 16
#define ECC_SMALL_MAGIC 0x45434353u
struct EccSmallEval {
    unsigned long long* done_out;   // pinned, device-mapped word the host polls: done_token once every value is on its way; null: no hand-over
    unsigned long long done_token;
    unsigned* ticket;    // device counter, zero between launches
    float* values_host;  // pinned, device-mapped array: every pair value, system scope (the host adds them)
    int stage_stride;    // floats per pair of the LDS stage (set by the launcher)
    unsigned long long* dbg;  // optional (experiments, -DECC_SMALL_STAMPS): 4 wall-clock stamps per workgroup
    unsigned magic;      // ECC_SMALL_MAGIC (set by the launcher): the kernel reads the patch list in place in its argument segment
    // E1 of the views whose matrix changed since the device arrays PinvTs / Cs were made, computed on the host
    // (ecc_host_geometry.h) and handed over in the kernel arguments: (P^+)^T (12) + C (4) per entry
    int patch_count;
    int patch_views[ECC_SMALL_PATCH_MAX];
    float patch_geo[ECC_SMALL_PATCH_MAX][16];
};

// ---- projection pre-processing ------------------------------------------------
#define ECC_PRE_MAX_CHUNKS 32  // workgroups per image of the maximum search in front of PreProccess::process (normalize)
struct EccPreprocessParams {
    const float* in;          // n_img images, n_v x n_u, u fastest
    float* out;               // same shape; must not alias `in` (tiles read halos of their neighbours)
    int64_t stride;           // floats between images
    int n_img, n_u, n_v;
    int process;              // 0: skip PreProccess::process, cosine weighting only
    int normalize;
    float scale, bias;
    float* max_d;             // n_img x ECC_PRE_MAX_CHUNKS partial maxima (normalize)
    int apply_log;
    int flip_u, flip_v;
    int zero[4], feather[4];  // left, right, bottom, top
    int n_blanks;
    const int* blanks;        // n_blanks x 4 on the device
    int k;                    // half kernel width of the low-pass, 0 = off
    const double* kernel;     // 2k+1 doubles on the device
    const float* cosw;        // n_img x 3 (sdd_px, ppu, ppv) or null
    const int* cosw_valid;    // n_img flags: 0 when the projection matrix is all zero (ref: PreProccess.cpp:149)
    // border factors by SOURCE position, 1.0f where the reference does not multiply (x * 1.0f == x bit for bit):
    // left[n_u], right[n_u], bottom[n_v], top[n_v], applied in this order (ref: Gui/PreProccess.cpp:86-113)
    const float* border_w;
};

// ---- MetricDirect -----------------------------------------------------------
struct EccDirectView {
    double C[4];     // source position, w = 1
    double Q[12];    // row-QR of P: orthonormal rows (3 x 4)
    double L[9];     // lower-triangular factor (3 x 3)
    double radius;   // estimateObjectRadius of this view
    double P[12];    // the projection matrix itself (column-major), for the FBCC weighting
};
struct EccDirectPair {
    double E0[4], E90[4];     // the two reference epipolar planes (Hessian normal form)
    double k_first, dkappa;   // kappa grid: k_first + dkappa * k
    int n_lines, i, j, pad;
    // FBCC only (ref: EpipolarConsistencyDirect.cpp:133-151): rectifying homographies (row-major 3 x 3),
    // baseline direction, virtual detector plane
    double H0[9], H1[9], dvec[3], Eplane[4];
};
struct EccDirectParams {
    const float* images;        // n_views images, n_v x n_u, u fastest
    const float* imagesT;       // the same images transposed (n_u x n_v, v fastest), or null
    int64_t image_stride;
    const EccDirectView* views;
    EccDirectPair* pairs;       // `count` records (direct_pair_kernel -> the other kernels)
    const int32_t* idx2;        // optional explicit (i, j) per pair; null = get_ij order from `first`
    float* samples;             // count x 2 x n_max line integrals
    double* pair_metric;        // count float64 pair values
    float* cost;                // optional n x n cost image (index i + j*n)
    int* pair_lines;            // optional: n_lines per pair
    float* debug_lines;         // optional: 6 floats per kappa of pair 0
    float* debug_kappas;        // optional: kappa grid of pair 0 (with debug_lines)
    int64_t first, count;
    int n_views, n_u, n_v, n_max;
    double object_radius_mm, dkappa;
    int use_fbcc;               // MetricDirect::setFanBeamConsistency
    const float* user_kappas;   // optional caller-provided kappa grid (ref: EpipolarConsistencyDirect.cpp:105-117), device
    int n_user_kappas;
};

// ---- evaluateForImagePair (E7, visualisation) -------------------------------------------------
struct EccPairSamplesParams {
    const float* dtr0;       // slab of view i
    const float* dtr1;       // slab of view j
    const float* Cs;
    const float* PinvTs;
    float* out;              // 7 x capacity floats: v0, v1, kappa, a0, d0, a1, d1
    float* K01_out;          // 16 floats
    int* n_out;              // number of samples produced (zeroed by the caller)
    int iP0, iP1;
    int capacity;
    int n_alpha, n_t, pitch;
    float n_x2, n_y2;
    float object_radius_mm;
    float num_samples;       // sqrtf(n_u*n_u + n_v*n_v)  (ref: ...RadonIntermediate.cpp:349)
    float range_t;
    float dkappa_user;
    int derivative0, derivative1;
};

#endif
