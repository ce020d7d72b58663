// preprocess_kernel.hip -- projection pre-processing on the device, the step in front of the Radon
// intermediate.
//
// The reference does this on the host, one image at a time, in five sweeps over the image plus two
// convolution passes (ref: LibEpipolarConsistency/Gui/PreProccess.cpp:57-166, HeaderOnly/NRRD/nrrd_lowpass.hxx:45-79)
// and then uploads the result for the Radon kernel (RadonIntermediate.cpp:27).  Here the whole chain is ONE
// kernel per batch, HBM-bound by construction (4 B read + 4 B written per pixel, halo re-reads come out of L2):
//   workgroup = 64 x 64 output pixels; it stages the (64+2k)^2 input footprint into LDS *through* the pixel-wise
//   stages (intensity -> border zero/feather -> blanks, evaluated at the un-flipped source position, i.e. the
//   flips are an index map), runs the horizontal Gaussian pass LDS -> LDS, the vertical pass LDS -> registers,
//   multiplies with the cosine weight and writes each pixel once.
// Arithmetic is the reference's, expression by expression (float pixel ops; the two convolution sums in binary64
// over o = -k .. k-1, the dropped last tap included; clamp addressing), so results are bit-identical to
// oracle/ecc_oracle.c (eccor_preprocess + eccor_cos_weight).
#include <hip/hip_runtime.h>
#include <math.h>

#include <algorithm>

#include "ecc_layout.h"

namespace {

constexpr int PP_TW = 64, PP_TH = 64;  // PP_TW = one wave per tile row

// ref: Gui/PreProccess.cpp:8-13 (weighting, double)
__device__ __forceinline__ double weighting_d(double x)
{
    if (x < -1.0 || x > 1.0) return 0;
    const double xx = x * x;
    return 1.0 - 2 * xx + xx * xx;
}

// ref: Gui/PreProccess.cpp:156-165
__device__ __forceinline__ float cos_weight(int gx, int gy, float ppu, float ppv, float sdd)
{
    const float pou = (float)gx - ppu;
    const float pov = (float)gy - ppv;
    return sdd / sqrtf(pou * pou + pov * pov + sdd * sdd);
}

// ref: Gui/PreProccess.cpp:78-84 (intensity map, optional -log, negative / non-finite values to zero)
__device__ __forceinline__ float intensity(const EccPreprocessParams& p, float v, float scale, float bias)
{
    float pixel = v * scale + bias;
    if (p.apply_log) pixel = (float)-(float)log((double)pixel);
    if (pixel < 0 || isnan(pixel) || isinf(pixel)) pixel = 0;
    return pixel;
}

// Pixel-wise part of PreProccess::process for the pixel at SOURCE position (sx, sy)
// (ref: Gui/PreProccess.cpp:78-120, same order: intensity, left, right, bottom, top, blanks).
__device__ __forceinline__ float blanked(const EccPreprocessParams& p, float pixel, int sx, int sy)
{
    for (int q = 0; q < p.n_blanks; ++q) {
        const int* bl = p.blanks + 4 * q;
        // ref: :116-119 (the y loop is bounded by img.size(0) there; clipped to the image as well)
        if (sx >= bl[0] && sx < bl[2] && sy >= bl[1] && sy < bl[3] && sy < p.n_u) pixel = 0;
    }
    return pixel;
}

__device__ __forceinline__ float borders(const EccPreprocessParams& p, float pixel, int sx, int sy)
{
    if (sx < p.zero[0] + p.feather[0])
        pixel *= sx <= p.zero[0] ? 0 : (float)weighting_d(1 - (float)(sx - p.zero[0]) / p.feather[0]);
    {
        const int b = p.n_u - sx;
        if (b <= p.zero[1] + p.feather[1])
            pixel *= b <= p.zero[1] ? 0 : (float)weighting_d(1 - (float)(b - p.zero[1]) / p.feather[1]);
    }
    {
        const int b = p.n_v - sy;
        if (b <= p.zero[2] + p.feather[2])
            pixel *= b <= p.zero[2] ? 0 : (float)weighting_d(1 - (float)(b - p.zero[2]) / p.feather[2]);
    }
    if (sy < p.zero[3] + p.feather[3])
        pixel *= sy <= p.zero[3] ? 0 : (float)weighting_d(1 - (float)(sy - p.zero[3]) / p.feather[3]);
    return blanked(p, pixel, sx, sy);
}

__device__ __forceinline__ float pointwise(const EccPreprocessParams& p, float v, int sx, int sy, float scale, float bias)
{
    return borders(p, intensity(p, v, scale, bias), sx, sy);
}

// Workgroup -> tile.  The launch is one-dimensional; workgroups b, b + 8, ... share an XCD (round-robin dispatch), and an
// XCD walks a CONTIGUOUS run of tiles (x fastest, then y, then the image), so the halo texels two neighbouring tiles both
// read come out of that XCD's L2.  With the three-dimensional grid of rounds 1-4 neighbouring tiles sat on different XCDs
// and every halo line was fetched again: 1.94x the image per launch through the fabric (profiles/r05_pmc_preprocess.txt).
// Returns false for the padding workgroups of the last round.
__device__ __forceinline__ bool pp_tile(const EccPreprocessParams& p, int& bx, int& by, int& bz)
{
    const unsigned tiles_x = (unsigned)(p.n_u + PP_TW - 1) / PP_TW, tiles_y = (unsigned)(p.n_v + PP_TH - 1) / PP_TH;
    const unsigned total = tiles_x * tiles_y * (unsigned)p.n_img, per_xcd = (total + 7) / 8;
    const unsigned t = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (t >= total) return false;
    const unsigned row = t / tiles_x;
    bx = (int)(t - row * tiles_x);
    bz = (int)(row / tiles_y);
    by = (int)(row - (unsigned)bz * tiles_y);
    return true;
}

// KT: half kernel width known at compile time (taps live in registers, loops unroll), or -1 = runtime k with
// (k = 0, no low-pass, is preprocess_pointwise_kernel's; the KT == 0 branches below are not instantiated any more)
// the taps read from an LDS copy (a scalar load per tap inside the loop serialises on its latency: 10.8 -> x us).
// NT: threads per workgroup -- 512 with the compile-time low-pass (more waves per CU for the same LDS: 5.2 -> 4.7 us per
// image; 1024: 5.9), 256 without a low-pass (3.2 against 3.5 us).
template <int KT, int NT>
__global__ __launch_bounds__(NT) void preprocess_kernel(EccPreprocessParams p)
{
    constexpr int PP_THREADS = NT;
    constexpr int PP_SEG = PP_TH / (NT / 64);  // outputs per thread along the direction of a convolution pass
    extern __shared__ float lds[];
    const int k = KT >= 0 ? KT : p.k;
    const int AW = PP_TW + 2 * k, AH = PP_TH + 2 * k;
    // odd row strides: both passes are conflict-free whether the lanes of a wave are 64 columns or 64 rows
    const int AS = AW | 1;
    constexpr int BS = PP_TW + 1;
    float* A = lds;            // AH x AW : pixel-wise result incl. halo (clamped = the convolution's clamp addressing)
    float* B = lds + AH * AS;  // AH x PP_TW : after the horizontal pass
    double* taps_lds = reinterpret_cast<double*>(lds + ((AH * AS + AH * BS + 1) & ~1));  // runtime-k path only (8-byte aligned)
    double taps[KT > 0 ? 2 * KT : 1];
    if (KT > 0) {
#pragma unroll
        for (int o = 0; o < 2 * KT; ++o) taps[o] = p.kernel[o];
    } else if (KT < 0) {
        if ((int)threadIdx.x < 2 * k) taps_lds[threadIdx.x] = p.kernel[threadIdx.x];
    }
    int bx, by, img_i;
    if (!pp_tile(p, bx, by, img_i)) return;  // (uniform over the workgroup, in front of every barrier)
    const float* __restrict__ src = p.in + (int64_t)img_i * p.stride;
    float* __restrict__ dst = p.out + (int64_t)img_i * p.stride;
    const int x0 = bx * PP_TW, y0 = by * PP_TH;
    const int W = p.n_u, H = p.n_v;

    float scale = p.scale, bias = p.bias;
    if (p.normalize) {  // ref: :68-76
        bias = 0;
        // the image's maximum from the partial maxima of image_max_kernel, with the reference's rule for NaNs
        // (ref: :68-71: max = img[0]; if (img[i] > max) max = img[i] -- NaNs never win, a NaN first pixel stays)
        float m = -INFINITY;
        for (int c = 0; c < ECC_PRE_MAX_CHUNKS; ++c) {
            const float v = p.max_d[img_i * ECC_PRE_MAX_CHUNKS + c];
            if (v > m) m = v;
        }
        const float first = src[0];
        scale = p.scale / (isnan(first) ? first : m);
    }

    // thread (tx, ty) walks rows ty, ty+4, ... and columns tx, tx+64, ...: no integer divisions, and the 64
    // lanes of a wave read 64 consecutive texels (reversed when flipped)
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (KT > 0) {
        // compile-time footprint: ALL of a thread's loads are issued before the first is consumed (a load -> pointwise ->
        // store loop waits one memory round trip per element, ~20 per thread): 9.8 -> 7.9 us per 1024^2 image with the
        // default low-pass.  (Without a low-pass the plain loop below is faster, 3.2 against 6.1 us.)
        // The footprint is walked as ONE flat index range: with rows of 64 + 2k texels per wave the second trip of each
        // row had 2k of 64 lanes active and the pixel-wise stage -- the most expensive part of this kernel, not the
        // convolution -- ran at half efficiency.
        constexpr int KK = KT > 0 ? KT : 1, AHc = PP_TH + 2 * KK, AWc = PP_TW + 2 * KK;
        constexpr int NE = (AHc * AWc + PP_THREADS - 1) / PP_THREADS;
        // pixel-wise stages other than the intensity map touch only pixels near the image border or inside a blank
        // rectangle; a workgroup whose footprint lies clear of them skips their tests (they are conditional in the
        // reference as well, so nothing changes for those pixels)
        bool interior = p.n_blanks == 0;
        {
            const int gx_lo = min(max(x0 - k, 0), W - 1), gx_hi = min(max(x0 + AWc - 1 - k, 0), W - 1);
            const int gy_lo = min(max(y0 - k, 0), H - 1), gy_hi = min(max(y0 + AHc - 1 - k, 0), H - 1);
            const int sx_lo = p.flip_u ? W - 1 - gx_hi : gx_lo, sx_hi = p.flip_u ? W - 1 - gx_lo : gx_hi;
            const int sy_lo = p.flip_v ? H - 1 - gy_hi : gy_lo, sy_hi = p.flip_v ? H - 1 - gy_lo : gy_hi;
            interior = interior && sx_lo >= p.zero[0] + p.feather[0] && p.n_u - sx_hi > p.zero[1] + p.feather[1] &&
                       p.n_v - sy_hi > p.zero[2] + p.feather[2] && sy_lo >= p.zero[3] + p.feather[3];
        }
        // border factors of this footprint's columns and rows (host-made tables, see EccPreprocessParams::border_w)
        float* T = reinterpret_cast<float*>(taps_lds);  // the runtime-k taps are not used on this path
        if (p.process && !interior) {
            for (int i = threadIdx.x; i < 2 * AWc + 2 * AHc; i += PP_THREADS) {
                const bool col = i < 2 * AWc;
                const int l = col ? (i < AWc ? i : i - AWc) : (i - 2 * AWc < AHc ? i - 2 * AWc : i - 2 * AWc - AHc);
                const int g = col ? min(max(x0 + l - k, 0), W - 1) : min(max(y0 + l - k, 0), H - 1);
                const int sp = col ? (p.flip_u ? W - 1 - g : g) : (p.flip_v ? H - 1 - g : g);
                const int table = col ? (i < AWc ? 0 : W) : (i - 2 * AWc < AHc ? 2 * W : 2 * W + H);
                T[i] = p.border_w[table + sp];
            }
        }
        float v[NE];
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            const int e = threadIdx.x + PP_THREADS * q;
            const int ly = e / AWc, lx = e - ly * AWc;
            const int gy = min(max(y0 + ly - k, 0), H - 1), gx = min(max(x0 + lx - k, 0), W - 1);
            const int sy = p.flip_v ? H - 1 - gy : gy, sx = p.flip_u ? W - 1 - gx : gx;
            v[q] = e < AHc * AWc ? src[(size_t)sy * W + sx] : 0.f;
        }
        // The unrolled part (registers -> LDS) carries only the intensity map; -log and the border / blank stages are
        // rolled loops over the thread's own LDS elements, entered only where they apply.  (With all of pointwise()
        // inlined 22 times -- two float64 logarithms each -- the kernel was 13 000 instructions, larger than the
        // instruction cache, and the pixel-wise stage cost 2 us per image on workgroups that skip almost all of it.)
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            const int e = threadIdx.x + PP_THREADS * q;
            const int ly = e / AWc, lx = e - ly * AWc;
            if (e < AHc * AWc) {
                float val = v[q];
                if (p.process) {
                    val = val * scale + bias;  // ref: Gui/PreProccess.cpp:78-84, continued below when apply_log
                    if (!p.apply_log && (val < 0 || isnan(val) || isinf(val))) val = 0;
                }
                A[ly * AS + lx] = val;
            }
        }
        if (p.process && !interior) __syncthreads();  // T (uniform per workgroup)
        if (p.process && (p.apply_log || !interior)) {
#pragma unroll 1
            for (int e = threadIdx.x; e < AHc * AWc; e += PP_THREADS) {
                const int ly = e / AWc, lx = e - ly * AWc;
                float val = A[ly * AS + lx];
                if (p.apply_log) {
                    val = (float)-(float)log((double)val);
                    if (val < 0 || isnan(val) || isinf(val)) val = 0;
                }
                if (!interior) {
                    val = val * T[lx];                   // left   (each product rounded to float, the reference's
                    val = val * T[AWc + lx];             // right    sequence of `pixel *= w`)
                    val = val * T[2 * AWc + ly];         // bottom
                    val = val * T[2 * AWc + AHc + ly];   // top
                    if (p.n_blanks) {
                        const int gy = min(max(y0 + ly - k, 0), H - 1), gx = min(max(x0 + lx - k, 0), W - 1);
                        val = blanked(p, val, p.flip_u ? W - 1 - gx : gx, p.flip_v ? H - 1 - gy : gy);
                    }
                }
                A[ly * AS + lx] = val;
            }
        }
    } else {
        for (int ly = ty; ly < AH; ly += PP_THREADS / 64) {
            const int gy = min(max(y0 + ly - k, 0), H - 1);
            const int sy = p.flip_v ? H - 1 - gy : gy;  // ref: :123-136
            for (int lx = tx; lx < AW; lx += 64) {
                const int gx = min(max(x0 + lx - k, 0), W - 1);
                const int sx = p.flip_u ? W - 1 - gx : gx;
                float v = src[(size_t)sy * W + sx];
                if (p.process) v = pointwise(p, v, sx, sy, scale, bias);
                A[ly * AS + lx] = v;
            }
        }
    }
    __syncthreads();
    if (k > 0) {
        // horizontal pass, ref: nrrd_lowpass.hxx:52-63 (o = -kx .. kx-1, double sum, result cast to T)
        if (KT > 0) {
            // A thread owns PP_SEG consecutive outputs of one row and walks their PP_SEG + 2k - 1 inputs once: each
            // input is read and widened to double ONCE and feeds up to 2k running sums (for a fixed output the taps
            // still arrive in the order o = 0 .. 2k-1, so the sums are the oracle's bit for bit).  One output per
            // thread widens every input 2k times, and v_cvt_f64_f32 is the slow instruction of this kernel.
            constexpr int KK = KT > 0 ? KT : 1, AHc = PP_TH + 2 * KK, NSEG = PP_TW / PP_SEG, NIN = PP_SEG + 2 * KK - 1;
            for (int i = threadIdx.x; i < AHc * NSEG; i += PP_THREADS) {
                const int row = i % AHc, seg = i / AHc;  // lanes = consecutive rows
                const float* a = A + row * AS + seg * PP_SEG;
                double sum[PP_SEG];
#pragma unroll
                for (int s = 0; s < PP_SEG; ++s) sum[s] = 0;
#pragma unroll
                for (int j = 0; j < NIN; ++j) {
                    const double d = a[j];
#pragma unroll
                    for (int s = 0; s < PP_SEG; ++s)
                        if (j - s >= 0 && j - s < 2 * KK) sum[s] += d * taps[j - s];
                }
#pragma unroll
                for (int s = 0; s < PP_SEG; ++s) B[row * BS + seg * PP_SEG + s] = (float)sum[s];
            }
        } else {
            for (int ly = ty; ly < AH; ly += PP_THREADS / 64) {
                double sum = 0;
                for (int o = 0; o < 2 * k; ++o) sum += A[ly * AS + tx + o] * taps_lds[o];
                B[ly * BS + tx] = (float)sum;
            }
        }
        __syncthreads();
    }
    const float sdd = p.cosw ? p.cosw[3 * img_i] : 0.f;
    const float ppu = p.cosw ? p.cosw[3 * img_i + 1] : 0.f, ppv = p.cosw ? p.cosw[3 * img_i + 2] : 0.f;
    const bool weight = p.cosw && p.cosw_valid[img_i];
    if (KT > 0) {
        // vertical pass, ref: nrrd_lowpass.hxx:64-75 (uses kernelx again, o = -ky .. ky-1): thread = column tx, rows
        // PP_SEG * ty .. + PP_SEG - 1, same single widening per input as above
        constexpr int KK = KT > 0 ? KT : 1, NIN = PP_SEG + 2 * KK - 1;
        static_assert(PP_SEG * (PP_THREADS / 64) == PP_TH, "one row segment per wave");
        const int lx = tx, gx = x0 + lx;
        const float* b = B + (PP_SEG * ty) * BS + lx;
        double sum[PP_SEG];
#pragma unroll
        for (int s = 0; s < PP_SEG; ++s) sum[s] = 0;
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
            const double d = b[j * BS];
#pragma unroll
            for (int s = 0; s < PP_SEG; ++s)
                if (j - s >= 0 && j - s < 2 * KK) sum[s] += d * taps[j - s];
        }
#pragma unroll
        for (int s = 0; s < PP_SEG; ++s) {
            const int gy = y0 + PP_SEG * ty + s;
            if (gx >= W || gy >= H) continue;
            float pixel = (float)sum[s];
            if (weight) pixel *= cos_weight(gx, gy, ppu, ppv, sdd);
            dst[(size_t)gy * W + gx] = pixel;
        }
        return;
    }
    for (int ly = ty; ly < PP_TH; ly += PP_THREADS / 64) {
        const int lx = tx;
        const int gx = x0 + lx, gy = y0 + ly;
        if (gx >= W || gy >= H) continue;
        float pixel;
        if (k > 0) {
            double sum = 0;
            for (int o = 0; o < 2 * k; ++o) sum += B[(ly + o) * BS + lx] * taps_lds[o];
            pixel = (float)sum;
        } else {
            pixel = A[ly * AS + lx];
        }
        if (weight) pixel *= cos_weight(gx, gy, ppu, ppv, sdd);
        dst[(size_t)gy * W + gx] = pixel;
    }
}

// Without a low-pass (sigma = 0 or half width <= 1, and the cosine-weighting-only call) nothing couples the pixels: no
// LDS, no barrier -- every thread loads its 16 pixels of a 64 x 64 tile first (all in flight), applies the pixel-wise
// stages (the border / blank tests only on tiles that touch a border zone or when blanks exist) and the cosine weight,
// and stores.  Same arithmetic as preprocess_kernel<0>, which it replaces: 3.3 -> 2.4 us per 1024^2 image.  (A thread owning
// 4 consecutive pixels of 4 rows with 16-byte loads and stores was measured: 3.1 us -- fewer loads in flight per wave.)
__global__ __launch_bounds__(256) void preprocess_pointwise_kernel(EccPreprocessParams p)
{
    constexpr int NP = PP_TH / 4;
    int bx, by, img_i;
    if (!pp_tile(p, bx, by, img_i)) return;
    const float* __restrict__ src = p.in + (int64_t)img_i * p.stride;
    float* __restrict__ dst = p.out + (int64_t)img_i * p.stride;
    const int W = p.n_u, H = p.n_v;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int gx = bx * PP_TW + tx, y0 = by * PP_TH;
    float scale = p.scale, bias = p.bias;
    if (p.normalize) {  // ref: :68-76, as in preprocess_kernel
        bias = 0;
        float m = -INFINITY;
        for (int c = 0; c < ECC_PRE_MAX_CHUNKS; ++c) {
            const float v = p.max_d[img_i * ECC_PRE_MAX_CHUNKS + c];
            if (v > m) m = v;
        }
        const float first = src[0];
        scale = p.scale / (isnan(first) ? first : m);
    }
    const int sx = p.flip_u ? W - 1 - gx : gx;
    bool interior = p.n_blanks == 0;
    {
        const int gx_lo = bx * PP_TW, gx_hi = min(gx_lo + PP_TW, W) - 1, gy_lo = y0, gy_hi = min(y0 + PP_TH, H) - 1;
        const int sx_lo = p.flip_u ? W - 1 - gx_hi : gx_lo, sx_hi = p.flip_u ? W - 1 - gx_lo : gx_hi;
        const int sy_lo = p.flip_v ? H - 1 - gy_hi : gy_lo, sy_hi = p.flip_v ? H - 1 - gy_lo : gy_hi;
        interior = interior && sx_lo >= p.zero[0] + p.feather[0] && p.n_u - sx_hi > p.zero[1] + p.feather[1] &&
                   p.n_v - sy_hi > p.zero[2] + p.feather[2] && sy_lo >= p.zero[3] + p.feather[3];
    }
    const float sdd = p.cosw ? p.cosw[3 * img_i] : 0.f;
    const float ppu = p.cosw ? p.cosw[3 * img_i + 1] : 0.f, ppv = p.cosw ? p.cosw[3 * img_i + 2] : 0.f;
    const bool weight = p.cosw && p.cosw_valid[img_i];
    if (p.process && !interior) {  // border tiles: the full pixel-wise chain in a rolled loop (long and rarely needed)
#pragma unroll 1
        for (int q = 0; q < NP; ++q) {
            const int gy = y0 + ty + 4 * q;
            if (gx >= W || gy >= H) continue;
            const int sy = p.flip_v ? H - 1 - gy : gy;
            float pixel = pointwise(p, src[(size_t)sy * W + sx], sx, sy, scale, bias);
            if (weight) pixel *= cos_weight(gx, gy, ppu, ppv, sdd);
            dst[(size_t)gy * W + gx] = pixel;
        }
        return;
    }
    float v[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int gy = y0 + ty + 4 * q;
        const int sy = p.flip_v ? H - 1 - gy : gy;
        v[q] = (gx < W && gy < H) ? src[(size_t)sy * W + sx] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int gy = y0 + ty + 4 * q;
        if (gx >= W || gy >= H) continue;
        float pixel = v[q];
        if (p.process) {
            pixel = pixel * scale + bias;  // ref: Gui/PreProccess.cpp:78-84
            if (p.apply_log) pixel = (float)-(float)log((double)pixel);
            if (pixel < 0 || isnan(pixel) || isinf(pixel)) pixel = 0;
        }
        if (weight) pixel *= cos_weight(gx, gy, ppu, ppv, sdd);
        dst[(size_t)gy * W + gx] = pixel;
    }
}

// Partial maxima for Intensity/Normalize (ref: Gui/PreProccess.cpp:68-71): ECC_PRE_MAX_CHUNKS workgroups per image, each
// the maximum of its contiguous part with `if (v > m) m = v` from -inf (NaNs never win); preprocess_kernel combines them.
// (One workgroup per image read its 4 MB alone: 9 us per image in a 50-image batch, 80 us for a single image.)
__global__ __launch_bounds__(1024) void image_max_kernel(const float* __restrict__ in, int64_t stride, int64_t len,
                                                         float* __restrict__ max_out)
{
    __shared__ float s[1024 / 64];
    const float* img = in + (int64_t)blockIdx.y * stride;
    const int64_t per = (len + ECC_PRE_MAX_CHUNKS - 1) / ECC_PRE_MAX_CHUNKS;
    const int64_t a = per * blockIdx.x, b = a + per < len ? a + per : len;
    float m = -INFINITY;
    for (int64_t i = a + threadIdx.x; i < b; i += 1024) {
        const float v = img[i];
        if (v > m) m = v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(m, off);
        if (o > m) m = o;
    }
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 1024 / 64; ++w)
            if (s[w] > m) m = s[w];
        max_out[(int64_t)blockIdx.y * ECC_PRE_MAX_CHUNKS + blockIdx.x] = m;
    }
}

}  // namespace

extern "C" size_t ecc_preprocess_lds_bytes(int k)
{
    const int AW = PP_TW + 2 * k, AH = PP_TH + 2 * k;
    const size_t floats = (size_t)AH * (AW | 1) + (k > 0 ? (size_t)AH * (PP_TW + 1) : 0);  // A and B with odd row strides
    // tail: the runtime-k taps (2k doubles) or the footprint's border factors (2 AW + 2 AH floats), whichever is larger
    return sizeof(float) * ((floats + 1) & ~(size_t)1) + std::max(sizeof(double) * 2 * (size_t)k, sizeof(float) * 2 * (size_t)(AW + AH));
}

extern "C" hipError_t ecc_launch_preprocess(const EccPreprocessParams* p, hipStream_t stream)
{
    if (p->normalize && p->process) {
        hipLaunchKernelGGL(image_max_kernel, dim3(ECC_PRE_MAX_CHUNKS, p->n_img), dim3(1024), 0, stream, p->in, p->stride,
                           (int64_t)p->n_u * p->n_v, p->max_d);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    // one-dimensional grid, padded to a multiple of 8: pp_tile gives every XCD a contiguous run of tiles
    const unsigned tiles = (unsigned)((p->n_u + PP_TW - 1) / PP_TW) * (unsigned)((p->n_v + PP_TH - 1) / PP_TH) * (unsigned)p->n_img;
    dim3 grid(((tiles + 7) / 8) * 8);
    const size_t lds = ecc_preprocess_lds_bytes(p->k);
    if (p->k == 0)
        hipLaunchKernelGGL(preprocess_pointwise_kernel, grid, dim3(256), 0, stream, *p);
    else if (p->k == 5)  // the reference's default (Gui/PreProccess.h:28)
        hipLaunchKernelGGL((preprocess_kernel<5, 512>), grid, dim3(512), lds, stream, *p);
    else
        hipLaunchKernelGGL((preprocess_kernel<-1, 256>), grid, dim3(256), lds, stream, *p);
    return hipGetLastError();
}
