// direct_kernel.hip -- MetricDirect: epipolar consistency straight from the projection images.
//
// The reference evaluates one pair at a time from the host (ref: LibEpipolarConsistency/EpipolarConsistencyDirect.cpp:
// 67-219,247-259): per pair it builds the kappa grid and the 2 x n_lines epipolar lines on the host in float64,
// uploads them, launches kernel_computeLineIntegrals twice (EpipolarConsistencyDirect.cu:31-125; 32-thread blocks,
// texture fetches, a device-wide sync after each), reads both signals back and sums on the host.  Here a batch of
// pairs is four stream-ordered launches with nothing crossing PCIe but the final scalar:
//   direct_view_kernel   thread per view: source position and the row-QR factors of P (float64)
//   direct_pair_kernel   thread per pair: baseline, the two reference planes, kappa range/step, line count
//   direct_lines_kernel  thread per (pair, image, kappa): the line in float64 -> float, then the reference's
//                        fp32 line integral (0.4-px steps, two parallel lines half a pixel apart) with the exact
//                        bilinear rule on the image in global memory; neighbouring threads are neighbouring lines
//                        of the pencil, so their taps share cache lines
//   direct_reduce_kernel wave per pair: sum (v0-v1)^2 dkappa in float64 (fixed order), cost image entry
// setFanBeamConsistency (RectifiedFBCC.h, ...Direct.cpp:133-196): the pair kernel also builds the two rectifying
// homographies, and each line thread derives its LinePerspectivity weighting in float64 before a weighted plain
// line integral -- the reference does that per line on the host.
// Arithmetic follows oracle/ecc_oracle.c (eccor_direct_pair): line integrals are bit-identical for identical
// lines; the lines themselves differ from the oracle's only through the float64 sin/cos of the two libms.
#include <hip/hip_runtime.h>
#include <math.h>

#include "ecc_host_geometry.h"
#include "ecc_layout.h"
#include "ecc_sampling.h"
#ifdef ECC_DIRECT_STATS  // scripts/direct_stats.py: slabs, steps through the tile / through global memory (radon_kernel.hip's scheme)
__device__ unsigned long long g_direct_stats[8];
#define ECC_SLAB_STAT(i, v) atomicAdd(&g_direct_stats[i], (unsigned long long)(v))
#endif
#include "ecc_slab_tile.h"

namespace {

__global__ __launch_bounds__(64) void direct_view_kernel(const double* __restrict__ Ps, int n, EccDirectView* __restrict__ views,
                                                         int n_u, int n_v)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    double P[12];
    for (int k = 0; k < 12; ++k) P[k] = Ps[12 * (size_t)v + k];
    EccDirectView out;
    ecc_host::camera_center(P, out.C);
    ecc_host::RowQR f;
    ecc_host::row_qr(P, &f);
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k < 4; ++k) out.Q[4 * i + k] = f.Q[i][k];
        for (int j = 0; j < 3; ++j) out.L[3 * i + j] = f.L[i][j];
    }
    out.radius = ecc_host::object_radius(P, n_u, n_v);
    for (int k = 0; k < 12; ++k) out.P[k] = P[k];
    views[v] = out;
}

// ref: EpipolarConsistencyDirect.cpp:84-117 + estimateAngularRange (EpipolarConsistency.cpp:49-59)
__global__ __launch_bounds__(256) void direct_pair_kernel(EccDirectParams p)
{
    const long long local = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (local >= p.count) return;
    int i, j;
    if (p.idx2) {
        i = p.idx2[2 * local];
        j = p.idx2[2 * local + 1];
    } else {
        ecc_get_ij_device(p.first + local, p.n_views, i, j);
    }
    const EccDirectView& V0 = p.views[i];
    const EccDirectView& V1 = p.views[j];
    EccDirectPair r;
    r.i = i;
    r.j = j;
    double B[6];
    ecc_host::join_points(V0.C, V1.C, B);
    double radius = p.object_radius_mm;
    if (radius <= 0) radius = V0.radius > V1.radius ? V0.radius : V1.radius;  // ref: :88-90
    const double Pi = 3.14159265358979323846264338327950288419716939937510582;
    const double mom = sqrt(B[3] * B[3] + B[1] * B[1] + B[0] * B[0]);
    const double dir = sqrt(B[2] * B[2] + B[4] * B[4] + B[5] * B[5]);
    const double baseline_dist = mom / dir;
    double k_first, k_second;
    if (baseline_dist <= radius) {
        k_first = -0.5 * Pi;
        k_second = 0.5 * Pi;
    } else {
        const double km = fabs(asin(radius / baseline_dist));
        k_first = -km;
        k_second = km;
    }
    double dkappa = p.dkappa;
    if (dkappa <= 0) {
        const double diag = sqrt((double)(p.n_u * p.n_u + p.n_v * p.n_v));
        dkappa = 0.5 * (k_second - k_first) / diag;
    }
    const double nl = (k_second - k_first) / dkappa;
    int n_lines = nl < 2147483000.0 ? (int)nl : 2147483000;
    if (!(nl >= 0)) n_lines = 0;            // NaN geometry (coincident source positions)
    if (p.user_kappas) n_lines = p.n_user_kappas;  // ref: :105-106: a non-empty `kappas` is taken as the grid
    if (n_lines > p.n_max) n_lines = p.n_max;  // cannot happen for the host's bound; keeps the stores in range
    r.k_first = k_first;
    r.dkappa = dkappa;
    r.n_lines = n_lines;
    const double origin3[4] = {0, 0, 0, 1};
    ecc_host::join_line_point(B, origin3, r.E0);
    ecc_host::join_line_point(B, r.E0, r.E90);
    const double n0 = sqrt(r.E0[0] * r.E0[0] + r.E0[1] * r.E0[1] + r.E0[2] * r.E0[2]);
    const double n90 = sqrt(r.E90[0] * r.E90[0] + r.E90[1] * r.E90[1] + r.E90[2] * r.E90[2]);
    for (int k = 0; k < 4; ++k) {
        r.E0[k] /= n0;
        r.E90[k] /= n90;
    }
    r.pad = 0;
    for (int k = 0; k < 9; ++k) r.H0[k] = r.H1[k] = 0;
    r.dvec[0] = -B[2]; r.dvec[1] = -B[4]; r.dvec[2] = -B[5];
    r.Eplane[0] = r.Eplane[1] = r.Eplane[2] = r.Eplane[3] = 0;
    if (p.use_fbcc) {
        // virtual detector plane spanned by the baseline direction and its moment, rectifying homographies
        // (ref: EpipolarConsistencyDirect.cpp:133-151)
        const double U[3] = {r.dvec[0] / dir, r.dvec[1] / dir, r.dvec[2] / dir};
        const double Vv[3] = {B[3] / mom, -B[1] / mom, B[0] / mom};
        ecc_host::cross3(U, Vv, r.Eplane);
        r.Eplane[3] = 0;
        ecc_host::RowQR f;
        for (int which = 0; which < 2; ++which) {
            const EccDirectView& W = which ? V1 : V0;
            for (int a = 0; a < 3; ++a) {
                for (int q = 0; q < 4; ++q) f.Q[a][q] = W.Q[4 * a + q];
                for (int b = 0; b < 3; ++b) f.L[a][b] = W.L[3 * a + b];
            }
            ecc_host::fbcc_homography(f, W.C, U, Vv, r.Eplane, which ? r.H1 : r.H0);
        }
    }
    p.pairs[local] = r;
}

// ---- line integrals (ref: EpipolarConsistencyDirect.cu:31-125, kernel_computeLineIntegrals) ---------------------
// One thread per epipolar line, stepping 0.4 px along the clipped line and accumulating bilinear samples sequentially
// in fp32 in the reference's order: the derivative form takes two samples half a pixel to either side of the line
// (sump, summ), the fan-beam form one weighted sample.  A workgroup is 256 ADJACENT lines of one pair's pencil in one
// image: they are nearly parallel (the epipole of a C-arm scan is far outside the image) and sweep a narrow band --
// exactly the Radon kernel's situation, so the band is walked slab by slab through the LDS tile of texel pairs of
// ecc_slab_tile.h instead of gathering from global memory (round 2: 64 lanes = 64 cache lines per load even with the
// transposed image copy).  Same arithmetic per sample (ecc_sampling.h), bit-identical results.
// The reference's launcher hands n_u over for both image sizes (:137); the evident intent (n_u, n_v) is implemented,
// identical for square images (oracle/ecc_oracle.c does the same).

// ref: EpipolarConsistencyDirect.cu:44-76 (line -> origin, direction, clipped parameter range)
__device__ __forceinline__ ecc_slab::LineRun direct_line_run(float l0, float l1, float l2, int n_u, int n_v)
{
    float o0 = -l2 * l0, o1 = -l2 * l1;
    const float d0 = l1, d1 = -l0;
    float ts[4] = {(1 - o0) / d0, (n_u - 1 - o0) / d0, (1 - o1) / d1, (n_v - 1 - o1) / d1};
    if ((double)(d0 * d0) < 1e-12) ts[0] = -(ts[1] = 1e10f);
    if ((double)(d1 * d1) < 1e-12) ts[2] = -(ts[3] = 1e10f);
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++)
            if (ts[i] > ts[i + 1]) {
                const float tmp = ts[i];
                ts[i] = ts[i + 1];
                ts[i + 1] = tmp;
            }
    const float t_min = ts[1], t_max = ts[2];
    const float u = o0 + t_min * d0, v = o1 + t_min * d1;
    const bool inside = (u <= n_u && v <= n_v && u >= 0 && v >= 0);  // else the reference returns 0 (:78-82)
    o0 += .5f;
    o1 += .5f;
    // a degenerate line (NaN) fails every loop condition; t_max - t_min <= the image diagonal otherwise
    return ecc_slab::LineRun{o0, o1, d0, d1, t_min, t_max, inside};
}

struct DirectDerivSums {  // ref: EpipolarConsistencyDirect.cu:103-113
    float sump = 0.f, summ = 0.f;
    __device__ __forceinline__ void add(float vA, float vB, float)
    {
        sump += vA * 0.4f;
        summ += vB * 0.4f;
    }
};
struct DirectFbccSum {  // ref: EpipolarConsistencyDirect.cu:87-101 (the fbcc_d branch): rectified by weighting, no derivative
    float sum = 0.f;
    ecc_host::FbccInfo fbcc;
    __device__ __forceinline__ void add(float vA, float, float t)
    {
        const float u_prime = fbcc.transform(t) - fbcc.t_prime_ak;
        const float fbcc_weight = fbcc.derivative(t) / sqrtf(u_prime * u_prime + fbcc.d_l_kappa_C_sq);
        sum += 0.4f * vA * fbcc_weight;
    }
};

// The band of a workgroup's lines from four workgroup-wide extremes: the smallest / largest fast coordinate of any line
// (both samples) at two reference slow coordinates.  Every line is linear in s, so between the references the band lies
// within the interpolated extremes -- whatever the lines are (no pencil assumption); tight when they do not cross
// inside the image.
struct LinearBand {
    float f0lo, f0hi, f1lo, f1hi, s0, inv_ds01, beta;
    __device__ __forceinline__ void operator()(float sa, float sb, float& gmin, float& gmax) const
    {
        const float la = (sa - s0) * inv_ds01, lb = (sb - s0) * inv_ds01;
        const float loa = f0lo + (f1lo - f0lo) * la - beta * sa, lob = f0lo + (f1lo - f0lo) * lb - beta * sb;
        const float hia = f0hi + (f1hi - f0hi) * la - beta * sa, hib = f0hi + (f1hi - f0hi) * lb - beta * sb;
        gmin = fminf(loa, lob);
        gmax = fmaxf(hia, hib);
    }
};

template <bool TRANSP>
__device__ __forceinline__ float direct_lines_body(const EccDirectParams& p, ecc_slab::Shared& sh, const float* img,
                                                   const float* imgT, ecc_slab::LineRun& ln, float lh0, float lh1, float sigma,
                                                   float beta0, const ecc_host::FbccInfo* fbcc)
{
    using namespace ecc_slab;
    const int W = p.n_u, H = p.n_v;
    const int Ns = TRANSP ? W : H;
    const float of = TRANSP ? ln.o1 : ln.o0, os = TRANSP ? ln.o0 : ln.o1;
    const float df = TRANSP ? ln.d1 : ln.d0, ds = TRANSP ? ln.d0 : ln.d1;
    const float hf = TRANSP ? lh1 : lh0, hs = TRANSP ? lh0 : lh1;  // the derivative pair's offset from the line
    BandFrame bf;
    bf.sigma = sigma;
    set_beta(bf, beta0);
    bf.marg = uni(.75f);
    const bool runs_with = ln.active && ds * sigma > .3f;  // (the walker's own rule)
    LinearBand band;
    band.s0 = -2.f;
    band.inv_ds01 = 1.f / ((float)Ns + 4.f);
    band.beta = bf.beta;
    {
        const float m = runs_with ? df / ds : 0.f;
        const float s1 = (float)Ns + 2.f;
        // the line itself at the two reference coordinates, widened by the derivative pair's offsets
        const float c = of - os * m;  // f = c + m s
        const float w = fbcc ? 0.f : fabsf(hf) + fabsf(hs * m);
        float lo0 = runs_with ? c + m * band.s0 - w : FLT_MAX, hi0 = runs_with ? c + m * band.s0 + w : -FLT_MAX;
        float lo1 = runs_with ? c + m * s1 - w : FLT_MAX, hi1 = runs_with ? c + m * s1 + w : -FLT_MAX;
        block_min_max(sh, lo0, hi0);
        block_min_max(sh, lo1, hi1);
        band.f0lo = lo0; band.f0hi = hi0; band.f1lo = lo1; band.f1hi = hi1;
    }
    const float* src = TRANSP ? imgT : img;
    if (fbcc) {
        DirectFbccSum acc;
        acc.fbcc = *fbcc;
        walk<TRANSP, false, false, false>(sh, img, W, H, src, bf, band, ln, 0.f, 0.f, 0.f, 0.f, 0.4f, acc);
        return acc.sum;
    }
    DirectDerivSums acc;
    walk<TRANSP, true, true, false>(sh, img, W, H, src, bf, band, ln, lh0, lh1, -lh0, -lh1, 0.4f, acc);
    return acc.sump - acc.summ;
}

// ref: EpipolarConsistencyDirect.cpp:113-117 (kappa grid, float) and :49-63 (the line of plane kappa), normalised
__device__ __forceinline__ void direct_line_of(const EccDirectParams& p, const EccDirectPair& r, const EccDirectView& V, int k,
                                               float& lf0, float& lf1, float& lf2, float& kf)
{
    kf = p.user_kappas ? p.user_kappas[k] : (float)(r.k_first + r.dkappa * k);
    const double kappa = kf, c = cos(kappa), s = sin(kappa);
    double E[4], l[3];
    for (int q = 0; q < 4; ++q) E[q] = c * r.E0[q] + s * r.E90[q];
    ecc_host::RowQR f;
    for (int a = 0; a < 3; ++a) {
        for (int q = 0; q < 4; ++q) f.Q[a][q] = V.Q[4 * a + q];
        for (int b = 0; b < 3; ++b) f.L[a][b] = V.L[3 * a + b];
    }
    ecc_host::plane_to_line(f, E, l);
    const double nn = sqrt(l[0] * l[0] + l[1] * l[1]);
    lf0 = (float)(l[0] / nn);
    lf1 = (float)(l[1] / nn);
    lf2 = (float)(l[2] / nn);
}

// Small images (either side below DIRECT_SLAB_MIN_SIZE pixels): 256 adjacent lines of a pencil are a third of such an
// image wide, the slabs of their band get short and wide, and the whole image is cache-resident anyway -- one thread per
// line gathering from global memory (round 1-2's kernel) is faster there (256^2: 0.33 against 0.52 ms per 28-pair
// evaluation; 512^2: 1.77 against 0.91).  The lanes of a wave are adjacent lines at the same step: for near-horizontal
// lines they differ in v, so those read the transposed copy.  Same arithmetic, same bits as the slab kernel.
constexpr int DIRECT_SLAB_MIN_SIZE = 384;
__global__ __launch_bounds__(256) void direct_lines_gather_kernel(EccDirectParams p)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const long long pair = blockIdx.y;
    const int which = blockIdx.z;
    const EccDirectPair& r = p.pairs[pair];
    if (k >= r.n_lines) return;
    const EccDirectView& V = p.views[which ? r.j : r.i];
    float lf0, lf1, lf2, kf;
    direct_line_of(p, r, V, k, lf0, lf1, lf2, kf);
    const bool use_T = p.imagesT && fabsf(lf1) > fabsf(lf0);
    const float* img = (use_T ? p.imagesT : p.images) + (int64_t)(which ? r.j : r.i) * p.image_stride;
    const int si = use_T ? p.n_v : 1, sj = use_T ? 1 : p.n_u;
    const ecc_slab::LineRun ln = direct_line_run(lf0, lf1, lf2, p.n_u, p.n_v);
    float v = 0.f;
    if (ln.active) {
        if (p.use_fbcc) {
            DirectFbccSum acc;
            const float lf[3] = {lf0, lf1, lf2};
            ecc_host::fbcc_line_info(V.P, V.C, which ? r.H1 : r.H0, r.dvec, r.Eplane, lf, &acc.fbcc);
            for (float t = ln.t; t <= ln.t_max; t += 0.4f)
                acc.add(ecc_tex_global_strided(img, p.n_u, p.n_v, si, sj, ln.o0 + t * ln.d0, ln.o1 + t * ln.d1), 0.f, t);
            v = acc.sum;
        } else {
            DirectDerivSums acc;
            const float h0 = 0.5f * lf0, h1 = 0.5f * lf1;
            for (float t = ln.t; t <= ln.t_max; t += 0.4f) {
                const float x = ln.o0 + t * ln.d0, y = ln.o1 + t * ln.d1;
                acc.add(ecc_tex_global_strided(img, p.n_u, p.n_v, si, sj, x + h0, y + h1),
                        ecc_tex_global_strided(img, p.n_u, p.n_v, si, sj, x - h0, y - h1), t);
            }
            v = acc.sump - acc.summ;
        }
    }
    p.samples[((size_t)pair * 2 + which) * p.n_max + k] = v;
    if (p.debug_lines && pair == 0) {
        float* dl = p.debug_lines + 6 * (size_t)k + 3 * which;
        dl[0] = lf0;
        dl[1] = lf1;
        dl[2] = lf2;
        if (which == 0) p.debug_kappas[k] = kf;
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void direct_lines_kernel(EccDirectParams p)
{
    __shared__ ecc_slab::Shared sh;
    __shared__ float s_first_line[2];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const long long pair = blockIdx.y;
    const int which = blockIdx.z;  // 0: first view of the pair, 1: second
    const EccDirectPair& r = p.pairs[pair];
    if ((int)(blockIdx.x * blockDim.x) >= r.n_lines) return;  // uniform: nothing below this line returns early
    const bool live = k < r.n_lines;
    const EccDirectView& V = p.views[which ? r.j : r.i];
    float lf0 = 0.f, lf1 = 1.f, lf2 = 0.f, kf = 0.f;
    if (live) direct_line_of(p, r, V, k, lf0, lf1, lf2, kf);
    ecc_slab::LineRun ln = direct_line_run(lf0, lf1, lf2, p.n_u, p.n_v);
    ln.active = ln.active && live;
    ecc_host::FbccInfo info;
    if (p.use_fbcc && live) {
        const float lf[3] = {lf0, lf1, lf2};
        ecc_host::fbcc_line_info(V.P, V.C, which ? r.H1 : r.H0, r.dvec, r.Eplane, lf, &info);
    }
    // The workgroup's tile orientation, walking direction and window slope come from its first line (uniform): the
    // tile's fast axis is the image axis the line NORMAL is closer to (as in the Radon kernel); lines with y-normals
    // stage from the transposed copy of the images.
    if (threadIdx.x == 0) {
        s_first_line[0] = lf0;
        s_first_line[1] = lf1;
    }
    __syncthreads();
    const float n0 = ecc_slab::uni(s_first_line[0]), n1 = ecc_slab::uni(s_first_line[1]);
    const float* img = p.images + (int64_t)(which ? r.j : r.i) * p.image_stride;
    const float* imgT = p.imagesT ? p.imagesT + (int64_t)(which ? r.j : r.i) * p.image_stride : nullptr;
    const bool transp = imgT && fabsf(n1) > fabsf(n0);
    // direction (d0, d1) = (l1, -l0); plain tile: s = y, ds = -l0, df = l1; transposed: s = x, ds = l1, df = -l0
    const float ds0 = transp ? n1 : -n0, df0 = transp ? -n0 : n1;
    const float sigma = ds0 >= 0.f ? 1.f : -1.f;
    const float beta0 = df0 / (fabsf(ds0) > 1e-6f ? ds0 : 1e-6f);
    float v;
    if (transp) v = direct_lines_body<true>(p, sh, img, imgT, ln, 0.5f * lf0, 0.5f * lf1, sigma, beta0, p.use_fbcc ? &info : nullptr);
    else v = direct_lines_body<false>(p, sh, img, imgT, ln, 0.5f * lf0, 0.5f * lf1, sigma, beta0, p.use_fbcc ? &info : nullptr);
    if (!live) return;
    if (!ln.active) v = 0.f;
    p.samples[((size_t)pair * 2 + which) * p.n_max + k] = v;
    if (p.debug_lines && pair == 0) {
        float* dl = p.debug_lines + 6 * (size_t)k + 3 * which;
        dl[0] = lf0;
        dl[1] = lf1;
        dl[2] = lf2;
        if (which == 0) p.debug_kappas[k] = kf;
    }
}

// ref: EpipolarConsistencyDirect.cpp:205-208 (metric += (v0-v1)*(v0-v1)*dkappa, float difference and square)
__global__ __launch_bounds__(256) void direct_reduce_kernel(EccDirectParams p)
{
    const int lane = threadIdx.x & 63;
    const long long pair = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= p.count) return;
    const EccDirectPair& r = p.pairs[pair];
    const float* v0 = p.samples + (size_t)pair * 2 * p.n_max;
    const float* v1 = v0 + p.n_max;
    double acc = 0.0;
    for (int k = lane; k < r.n_lines; k += 64) {
        const float d = v0[k] - v1[k];
        acc += (double)(d * d) * r.dkappa;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0) {
        p.pair_metric[pair] = acc;
        if (p.cost) p.cost[(size_t)r.i + (size_t)r.j * p.n_views] = (float)acc;  // ref: :255 cost_image.pixel(i,j)
        if (p.pair_lines) p.pair_lines[pair] = r.n_lines;
    }
}

// Sum of `count` float64 pair metrics in a fixed order, added to *total (ref: :247-259, cost += ecc).
__global__ __launch_bounds__(1024) void direct_sum_kernel(const double* __restrict__ vals, long long count,
                                                          double* __restrict__ total)
{
    __shared__ double s[1024 / 64];
    double acc = 0.0;
    for (long long k = threadIdx.x; k < count; k += 1024) acc += vals[k];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < 1024 / 64; w++) tot += s[w];
        *total += tot;
    }
}

// Transposed copies of the images (u becomes the slow axis), tiled through LDS.
__global__ void direct_transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int W, int H, int64_t stride)
{
    __shared__ float tl[32][33];
    const float* s = src + (int64_t)blockIdx.z * stride;
    float* d = dst + (int64_t)blockIdx.z * stride;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int x = x0 + threadIdx.x, y = y0 + r;
        tl[r][threadIdx.x] = (x < W && y < H) ? s[(size_t)y * W + x] : 0.f;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int x = x0 + r, y = y0 + threadIdx.x;
        if (x < W && y < H) d[(size_t)x * H + y] = tl[threadIdx.x][r];
    }
}

}  // namespace

extern "C" hipError_t ecc_launch_direct_transpose(const float* src, float* dst, int n, int W, int H, hipStream_t stream)
{
    dim3 grid((W + 31) / 32, (H + 31) / 32, n), block(32, 8);
    hipLaunchKernelGGL(direct_transpose_kernel, grid, block, 0, stream, src, dst, W, H, (int64_t)W * H);
    return hipGetLastError();
}

extern "C" hipError_t ecc_launch_direct_views(const double* Ps_d, int n, EccDirectView* views, int n_u, int n_v,
                                              hipStream_t stream)
{
    hipLaunchKernelGGL(direct_view_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, Ps_d, n, views, n_u, n_v);
    return hipGetLastError();
}

// One batch: p->count pairs starting at p->first (get_ij order); p->count <= 65535.
extern "C" hipError_t ecc_launch_direct_batch(const EccDirectParams* p, double* total, hipStream_t stream)
{
    if (p->count <= 0) return hipSuccess;
    hipLaunchKernelGGL(direct_pair_kernel, dim3((unsigned)((p->count + 255) / 256)), dim3(256), 0, stream, *p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    dim3 grid((p->n_max + 255) / 256, (unsigned)p->count, 2);
    if (p->n_u < DIRECT_SLAB_MIN_SIZE || p->n_v < DIRECT_SLAB_MIN_SIZE)
        hipLaunchKernelGGL(direct_lines_gather_kernel, grid, dim3(256), 0, stream, *p);
    else
        hipLaunchKernelGGL(direct_lines_kernel, grid, dim3(256), 0, stream, *p);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(direct_reduce_kernel, dim3((unsigned)((p->count + 3) / 4)), dim3(256), 0, stream, *p);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (total) {
        hipLaunchKernelGGL(direct_sum_kernel, dim3(1), dim3(1024), 0, stream, p->pair_metric, p->count, total);
        e = hipGetLastError();
    }
    return e;
}

#ifdef ECC_DIRECT_STATS
extern "C" __attribute__((visibility("default"))) void ecc_debug_direct_stats(unsigned long long* out, int reset)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_direct_stats), sizeof(unsigned long long) * 8);
    if (reset) {
        const unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_direct_stats), z, sizeof(z));
    }
}
#endif
