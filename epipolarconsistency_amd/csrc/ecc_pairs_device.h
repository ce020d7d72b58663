// ecc_pairs_device.h -- device code shared by the pair kernels (pairs_kernel.hip: k01_kernel, pairs_kernel,
// pairs_reference_kernel, pair_samples_kernel) and the one-launch evaluation of small pair sets (small_eval_kernel.hip):
// pair geometry (ref: EpipolarConsistencyCommon.hxx:82-149), the sampling loops of one pair (ref:
// EpipolarConsistencyRadonIntermediate.cu:71-113,214-276), the polynomial fit of the sample coordinates and the plain
// CPU-path arithmetic of ECC_SAMPLING_REFERENCE.  Everything is in an anonymous namespace: each translation unit gets
// its own copy, the arithmetic is the same by construction.
#ifndef ECC_PAIRS_DEVICE_H
#define ECC_PAIRS_DEVICE_H

#include <hip/hip_runtime.h>
#include <float.h>

#include "ecc_layout.h"

namespace {

constexpr int PK_THREADS = 256;
constexpr int PK_MAIN_WAVES = 4;  // waves (= pairs) per workgroup of pairs_kernel, wave w taking the pair w * nblk + block.  Measured
                                  // on the benchmark's launch: 1 wave 0.399 ms, 2 0.333, 4 0.326, 8 0.377, 16 0.408
                                  // (round 5, at seven waves per SIMD: 1 wave 0.389, 2 0.320 but a slower step, 4 0.318-0.323, 8 0.344)
constexpr int PK_MAIN_THREADS = 64 * PK_MAIN_WAVES;

// ref: EpipolarConsistencyCommon.hxx:82-90 (shiftOriginAndNormlaize)
__device__ __forceinline__ void shift_origin_and_normalize(float x, float y, float* Ki)
{
    Ki[2] += x * Ki[0] + y * Ki[1];
    Ki[5] += x * Ki[3] + y * Ki[4];
    float s0 = sqrtf(Ki[0] * Ki[0] + Ki[1] * Ki[1]);
#pragma unroll
    for (int i = 0; i < 6; i++) Ki[i] /= s0;
}

// ref: EpipolarConsistencyCommon.hxx:93-149 (computeK01), same expressions in fp32, in three parts so that a thread
// that needs one view's half computes only that (k01_kernel); compute_K01 below is the whole function.
// Part 1 (:115-129): Pluecker baseline B = C0 ^ C1, its norms and the pencil K = [E0, E90] of epipolar planes.
__device__ __forceinline__ void baseline_pencil(const float* __restrict__ C0, const float* __restrict__ C1, float* K, float& s2,
                                                float& s3)
{
    float B01 = C0[0] * C1[1] - C0[1] * C1[0];
    float B02 = C0[0] * C1[2] - C0[2] * C1[0];
    float B03 = C0[0] * C1[3] - C0[3] * C1[0];
    float B12 = C0[1] * C1[2] - C0[2] * C1[1];
    float B13 = C0[1] * C1[3] - C0[3] * C1[1];
    float B23 = C0[2] * C1[3] - C0[3] * C1[2];
    s2 = sqrtf(B12 * B12 + B02 * B02 + B01 * B01);
    s3 = sqrtf(B03 * B03 + B13 * B13 + B23 * B23);
    K[0] = +B12 / s2; K[1] = -B02 / s2; K[2] = +B01 / s2; K[3] = 0;
    K[4] = (-B01 * B13 - B02 * B23) / (s2 * s3);
    K[5] = (+B01 * B03 - B12 * B23) / (s2 * s3);
    K[6] = (+B02 * B03 + B12 * B13) / (s2 * s3);
    K[7] = -s2 / s3;
}

// Part 2 (:131-135): one view's 3x2 map kappa -> epipolar line, origin at the image centre, unit normal at kappa = 0.
__device__ __forceinline__ void project_pencil(const float* __restrict__ PinvT, const float* K, float n_x2, float n_y2, float* Kv)
{
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            float sum = 0;
#pragma unroll
            for (int s = 0; s < 4; s++) sum += PinvT[s * 3 + i] * K[j * 4 + s];
            Kv[j * 3 + i] = sum;
        }
    shift_origin_and_normalize(n_x2, n_y2, Kv);
}

// Part 3 (:137-148): baseline distance, view angle, kappa range and step.
__device__ __forceinline__ void pencil_range(float s2, float s3, float object_radius_mm, float num_samples, float dkappa,
                                             bool want_view_angle, float& K06, float& K07, float& K16, float& K17)
{
    K06 = s2 / s3;
    // Elementary functions correctly rounded (binary64, rounded once), like the oracle; they run once
    // per pair.  K0[7] (angle between the views) is not used by the metric: debug output only.
    K07 = want_view_angle ? -2.0f * (float)atan2((double)(-0.5f * s3), (double)(s2 / s3)) : 0.f;
    const float Pi = 3.14159265359f;
    if (K06 <= object_radius_mm) K17 = 0.5f * Pi;
    else K17 = (float)asin((double)(object_radius_mm / K06));
    if (dkappa <= 0.f) K16 = 2.f * K17 / num_samples;
    else K16 = dkappa;
}

__device__ void compute_K01(float n_x2, float n_y2, const float* __restrict__ C0, const float* __restrict__ C1,
                            const float* __restrict__ P0invT, const float* __restrict__ P1invT,
                            float object_radius_mm, float num_samples, float dkappa, bool want_view_angle,
                            float* K0, float* K1)
{
    float K[8], s2, s3;
    baseline_pencil(C0, C1, K, s2, s3);
    project_pencil(P0invT, K, n_x2, n_y2, K0);
    project_pencil(P1invT, K, n_x2, n_y2, K1);
    pencil_range(s2, s3, object_radius_mm, num_samples, dkappa, want_view_angle, K0[6], K0[7], K1[6], K1[7]);
}

// =================================================================================================
// The pair kernel.  Same algorithm as the reference, restructured for CDNA4:
//   * one WAVE per pair (4 pairs per 256-thread workgroup, no barrier, no LDS): 64 lanes x 23
//     iterations cover N_kappa = 1448 with 98 % lane utilisation; the pair's record comes from k01_kernel by
//     scalar loads and lives in SGPRs;
//   * FAST PATH (record.poly_ok, the normal case): the four sample positions of a kappa step come from the
//     record's polynomials (kappa_loop_poly / poly_pm), then one 16-byte load and the bilinear rule each;
//   * EXACT PATH (kappa_loop / sample_line), for pairs whose fit was rejected: per sample
//       - +kappa and -kappa share the six products K[:,0]*cos, K[:,1]*sin (x(-kappa) = (-cos, sin));
//       - the (alpha+pi, -t) periodicity fold is a sign-bit operation on the line instead of the
//         reference's atan2 range tests: a line with l1 < 0 is negated (same point set), which maps
//         a -> a-1, d -> 1-d, and the sample gets the sign bit back (derivative filter only);
//       - 1/len by v_rsq_f32, the angle by one v_rcp_f32 + a minimax polynomial of atan(q)/(pi q) in q^2
//         (degree 8 on [0,1]: max error 6.7e-8 in a, below the fp32 libm path of the oracle measured against
//         float64; degree 3 when the whole wave is near-horizontal), sin/cos(kappa) by pi/4-reduced kernels
//         (< 0.9 ulp); no IEEE division sequences in the loop.
// Differences to the oracle are at the ulp level of the sample coordinates; tests hold the mean
// to 1e-5 and pair values to 2e-4 (fp32 noise floor, tests/test_oracle_properties.py).
// =================================================================================================

struct __attribute__((packed, aligned(4))) F2 {
    float x, y;
};
// One bilinear footprint of the row-paired copy (see build_paired_kernel): (i,j), (i+1,j), (i,j+1), (i+1,j+1).
struct __attribute__((packed, aligned(4))) F4 {
    float x, y, z, w;
};

__device__ __forceinline__ float uniformf(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// sin and cos of kappa in [0, pi/2]: reduce to [0, pi/4] by kappa -> pi/2 - kappa (exact subtraction
// of the high part, Sterbenz), then degree-7 / degree-8 kernels.  REDUCE = false when the whole pair stays
// below pi/4 (kappa_max <= pi/4, wave-uniform): no reduction, no selects.
template <bool REDUCE>
__device__ __forceinline__ void sincos_quadrant(float kappa, float& s, float& c)
{
    const float pio2_hi = 1.57079637050628662109375f, pio2_lo = -4.37113900018624283e-8f;
    const bool swap = REDUCE && kappa > 0.785398163397448f;
    const float r = swap ? (pio2_hi - kappa) + pio2_lo : kappa;
    const float z = r * r;
    float ps = fmaf(fmaf(-1.958291686605662e-04f, z, 8.332724682986736e-03f), z, -1.66666641831398e-01f);
    float sn = fmaf(ps, r * z, r);
    float pc = fmaf(fmaf(fmaf(2.445712743792683e-05f, z, -1.3887537643313408e-03f), z, 4.166664928197861e-02f), z, -0.5f);
    float cs = fmaf(pc, z, 1.0f);
    s = swap ? cs : sn;
    c = swap ? sn : cs;
}

// The copies the pair kernel samples live in GLOBAL memory, and the pointer type says so: their base comes out of a device
// table (p.dtrs[i]), which leaves a plain pointer's address space unknown to the compiler -- flat loads with 64-bit vector
// address arithmetic instead of global_load saddr + 32-bit voffset (round 4: +6 % vector instructions, +9 % kernel time
// when the sampling code moved behind a reference parameter and the kernel-argument promotion no longer saw through it).
typedef const __attribute__((address_space(1))) char* GlobalBytes;
typedef const __attribute__((address_space(1))) float* GlobalFloats;
// one bilinear footprint (16 bytes, 4-byte aligned) as a built-in vector: loadable through a global-address-space pointer in
// the host pass of the compiler too (a struct would need a constructor from that address space)
typedef float ecc_v4f_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef const __attribute__((address_space(1))) ecc_v4f_a4* GlobalF4;
struct SlabView {
    GlobalBytes origin;  // base of the dtr's ROW-PAIRED copy: padded element (row 0, column 0) = (ix = -1, iy = -1)
    unsigned pitch4;     // row pitch of the paired copy in bytes (8 bytes per distance bin)
};

// atan(t)/pi for |t| <= 1 as t * P(t^2): degree-8 minimax fit of atan(q)/(pi q) on q^2 in [0,1]
// (max error 6.7e-8 in the angle a, below the oracle's own fp32 rounding of a).
__device__ __forceinline__ float atan_over_pi(float t)
{
    const float z = t * t;
    float pz = 9.021107107e-04f;
    pz = fmaf(pz, z, -5.094559398e-03f);
    pz = fmaf(pz, z, 1.355605666e-02f);
    pz = fmaf(pz, z, -2.385874465e-02f);
    pz = fmaf(pz, z, 3.385784104e-02f);
    pz = fmaf(pz, z, -4.520818591e-02f);
    pz = fmaf(pz, z, 6.363805383e-02f);
    pz = fmaf(pz, z, -1.061024442e-01f);
    pz = fmaf(pz, z, 3.183098733e-01f);
    return pz * t;
}

// Same for |t| <= 0.3 (lines within 16.7 deg of the image x axis): degree 3 is enough (max error 1.3e-8).
__device__ __forceinline__ float atan_over_pi_small(float t)
{
    const float z = t * t;
    float pz = -3.970951959e-02f;
    pz = fmaf(pz, z, 6.334859133e-02f);
    pz = fmaf(pz, z, -1.060977504e-01f);
    pz = fmaf(pz, z, 3.183098733e-01f);
    return pz * t;
}

// Byte offset of the footprint (row = floor(xa), bin = floor(yd), both already integral floats) in a row-paired copy.
// PITCH4 >= 0: formed exactly in fp32 -- one fma, one conversion -- which needs the copy to stay below 2^24 bytes
// (768 x 768 bins: 4.9 MB); PITCH4 > 0 is that with the pitch as a compile-time constant.  PITCH4 < 0: integer
// arithmetic (two conversions, one 24-bit multiply-add) for Radon intermediates of any size up to 16384 x 16384 bins
// (offsets < 2^32); chosen per launch by the host (EccPairParams::wide_offsets), ~5 % slower.
// PITCH4 == ECC_QUAD_LAYOUT: the offset in a ROW-QUAD copy (build_quad_kernel), where one 128-byte line holds the
// footprints of 4 consecutive rows x 2 bins; `pitch4` is then the byte size of one group of four rows.
constexpr int ECC_QUAD_LAYOUT = -2;
template <int PITCH4>
__device__ __forceinline__ unsigned footprint_offset(float row_f, float bin_f, unsigned pitch4, float pitch4_f)
{
    if (PITCH4 >= 0) return (unsigned)fmaf(row_f, PITCH4 > 0 ? (float)PITCH4 : pitch4_f, bin_f * 8.0f);
    if (PITCH4 == ECC_QUAD_LAYOUT) {
        const unsigned r = (unsigned)row_f;
        return __umul24(r >> 2, pitch4) + ((unsigned)bin_f << 6) + ((r & 3u) << 4);
    }
    return __umul24((unsigned)row_f, pitch4) + ((unsigned)bin_f << 3);
}

// One line (l0, l1, l2) -> signed bilinear sample of the dtr.
// Instruction selection follows measured gfx950 issue costs (scripts/micro/valu_rate.hip):
// v_fma/v_add/v_xor 2 cycles per wave64, v_floor/v_fract/v_cvt 4, v_cmp+v_cndmask 8 per pair,
// v_rcp/v_rsq 8.6 -- so selects are replaced by sign-bit arithmetic and a wave-uniform branch.
// PITCH4 > 0: row pitch in bytes known at
// compile time (the 768-bin default), the second row's load then uses an immediate offset.
struct LineTap {
    GlobalF4 ptr;  // the 2x2 footprint in the row-paired copy
    float fx, fy;   // bilinear weights
    unsigned m;     // sign bit of the fold
};

template <int PITCH4>
__device__ __forceinline__ LineTap sample_line_prep(float l0, float l1, float l2, const SlabView sv, float n_alpha_f,
                                                    float n_t_f, float dist_scale, float dist_bias, float pitch4_f)
{
    // fold: a line whose normal points to negative y is negated (the reference's a > 1 branch).  l0/l1 is
    // invariant under the negation, so only the distance term and the sample take the sign.
    const unsigned m = __float_as_uint(l1) & 0x80000000u;
    const float inv = __builtin_amdgcn_rsqf(fmaf(l0, l0, l1 * l1));
    const float l2f = __uint_as_float(__float_as_uint(l2) ^ m);
    const float al1 = fabsf(l1);

    // r = angle(l0, l1) / pi in [0, 1] as cq + u, cq in {0, 1/2, 1}, |u| <= 1/4 (for the folded line, l1 >= 0).
    float cq, u;
    const float fl0 = __uint_as_float(__float_as_uint(l0) ^ m);
    const float ax = fabsf(l0);
    if (__builtin_amdgcn_ballot_w64(ax > 0.3f * al1) == 0) {
        // every lane of the wave has a line within 16.7 deg of the image x axis (normal close to y):
        // r = 1/2 - atan(l0 / l1) / pi with the short polynomial.  This is the case almost all pairs of a
        // circular C-arm scan are in.  (Deciding it once per pair from a bound on K instead of per sample
        // was tried -- as a separate branch-free loop, 0.57 ms, and as a scalar flag, no change from this
        // form's 0.49 ms: the test and the ballot are cheap, and the branches give the scheduler block
        // boundaries that keep each sample's loads ahead of the next sample's arithmetic.)
        cq = 0.5f;
        u = -atan_over_pi_small(fl0 * __builtin_amdgcn_rcpf(al1));
    } else if (__builtin_amdgcn_ballot_w64(al1 < ax) == 0) {
        cq = 0.5f;  // within 45 deg
        u = -atan_over_pi(fl0 * __builtin_amdgcn_rcpf(al1));
    } else {
        const bool steep = al1 >= ax;
        const float num = steep ? fl0 : al1, den = steep ? al1 : fl0;  // signed quotient, |num/den| <= 1
        const float p = atan_over_pi(num * __builtin_amdgcn_rcpf(den));
        cq = steep ? 0.5f : (fl0 < 0.f ? 1.0f : 0.0f);
        u = steep ? -p : p;
    }
    // The reference divides by the FLOAT constant Pi = 3.14159265359f = pi (1 + e), e = 2.78e-8
    // (ref: EpipolarConsistencyCommon.hxx:155,159): a = r (1 - e) on the direct branch and
    // a = r + e (1 - r) on the folded (+2, -1) branch.  That 1e-5-bin shift of the sampling angle moves
    // the 400-view metric by 1.7e-5 (profiles/r01_parity_probes.txt), so it is reproduced here; the
    // correction is added to the small term u BEFORE the one rounding against cq, otherwise it would
    // vanish below half an ulp of a.  folded ? 1 : 0 = 0.5 * bitcast(m >> 1) (0x40000000 = 2.0f).
    const float e = 2.7827534e-8f;
    const float f_minus_r = fmaf(__uint_as_float(m >> 1), 0.5f, -(cq + u));
    const float a = cq + fmaf(f_minus_r, e, u);

    // texel position a*n_alpha - .5, d*n_t - .5 (normalised coordinates), expressed
    // directly in the slab's padded coordinates (+1): the replicated border stands in for clamp
    // addressing and all byte offsets are non-negative (saddr + 32-bit voffset loads).
    // d*n_t + .5 = (-(l2/len)/range_t + .5)*n_t + .5 in one fma: dist_scale = n_t/range_t, dist_bias = .5 n_t + .5.
    const float xa = fmaf(a, n_alpha_f, 0.5f);
    float yd = fmaf(-(l2f * inv), dist_scale, dist_bias);
    yd = __builtin_amdgcn_fmed3f(yd, 0.f, n_t_f);
    const float fx = __builtin_amdgcn_fractf(xa), fy = __builtin_amdgcn_fractf(yd);
    // byte offset floor(xa)*pitch4 + floor(yd)*8 in the row-paired copy; ONE 16-byte load fetches the whole 2x2 footprint
    const unsigned off = footprint_offset<PITCH4>(xa - fx, yd - fy, sv.pitch4, pitch4_f);
    LineTap t;
    t.ptr = (GlobalF4)(sv.origin + off);
    t.fx = fx;
    t.fy = fy;
    t.m = m;
    return t;
}

template <bool DERIV>
__device__ __forceinline__ float line_tap_finish(const F4 q, const LineTap t)
{
    const float r0 = fmaf(t.fx, q.y, q.x);  // q.y, q.w: the row differences, formed when the copy is built
    const float r1 = fmaf(t.fx, q.w, q.z);
    const float v = fmaf(t.fy, r1 - r0, r0);
    return DERIV ? __uint_as_float(__float_as_uint(v) ^ t.m) : v;
}

template <bool DERIV, int PITCH4>
__device__ __forceinline__ float sample_line(float l0, float l1, float l2, const SlabView sv, float n_alpha_f,
                                             float n_t_f, float dist_scale, float dist_bias, float pitch4_f)
{
    const LineTap t = sample_line_prep<PITCH4>(l0, l1, l2, sv, n_alpha_f, n_t_f, dist_scale, dist_bias, pitch4_f);
    // (non-temporal gathers for the kappa_max > pi/4 pairs, whose lines nobody re-uses, were measured inside the
    // benchmark's mixed launch: 0.344 vs 0.333 ms -- they lose their own L1 hits)
    const ecc_v4f_a4 q4 = *t.ptr;
    const F4 q = {q4.x, q4.y, q4.z, q4.w};
    return line_tap_finish<DERIV>(q, t);
}

// One kappa sample (four bilinear samples) of the pair loop; returns false when kappa is past kappa_max.
// WPP > 1 (several waves share the pair, small_eval_kernel.hip): the sample's term goes to stage[k] instead of acc; one wave
// adds the terms afterwards in the order a single wave accumulates them.
template <bool DERIV, bool CORR, bool REDUCE, int PITCH4, int WPP = 1>
__device__ __forceinline__ bool kappa_step(int k, const float (&K0)[8], const float (&K1)[8], float dkappa, float kappa_max,
                                           const SlabView sv0, const SlabView sv1, float n_alpha_f, float n_t_f,
                                           float dist_scale, float dist_bias, float pitch4_f, double& acc, double& mom2,
                                           double& mom3, double& mom4, float* stage = nullptr)
{
    static_assert(WPP == 1 || !CORR, "the shared-pair form carries one term per sample");
    const float kappa = dkappa * 0.5f + dkappa * k;  // ref: ...RadonIntermediate.cu:259 (same fp32 ops)
    if (kappa >= kappa_max) return false;
    float sn, cs;
    sincos_quadrant<REDUCE>(kappa, sn, cs);
    // view 0
    const float a00 = K0[0] * cs, a01 = K0[1] * cs, a02 = K0[2] * cs;
    const float b00 = K0[3] * sn, b01 = K0[4] * sn, b02 = K0[5] * sn;
    // view 1
    const float a10 = K1[0] * cs, a11 = K1[1] * cs, a12 = K1[2] * cs;
    const float b10 = K1[3] * sn, b11 = K1[4] * sn, b12 = K1[5] * sn;
    const float v0p = sample_line<DERIV, PITCH4>(b00 + a00, b01 + a01, b02 + a02, sv0, n_alpha_f, n_t_f, dist_scale, dist_bias, pitch4_f);
    const float v1p = sample_line<DERIV, PITCH4>(b10 + a10, b11 + a11, b12 + a12, sv1, n_alpha_f, n_t_f, dist_scale, dist_bias, pitch4_f);
    const float v0m = sample_line<DERIV, PITCH4>(b00 - a00, b01 - a01, b02 - a02, sv0, n_alpha_f, n_t_f, dist_scale, dist_bias, pitch4_f);
    const float v1m = sample_line<DERIV, PITCH4>(b10 - a10, b11 - a11, b12 - a12, sv1, n_alpha_f, n_t_f, dist_scale, dist_bias, pitch4_f);
    if (!CORR) {
        const float vp = v0p - v1p, vm = v0m - v1m;
        const float consistency = (vp * vp + vm * vm) * K0[6];  // ref: ...RadonIntermediate.cu:112
        if (WPP == 1) acc += (double)(consistency * dkappa);    // ref: ...RadonIntermediate.cu:269
        else stage[k] = consistency * dkappa;
    } else {
        // ref: ...RadonIntermediate.cu:116-149; the launcher passes kappa_max/kappa as "1/n" (:211,274)
        const float one_over_n = kappa_max / kappa;
        mom2 += (double)(one_over_n * (v0p * v0p + v0m * v0m));
        mom3 += (double)(one_over_n * (v1p * v1p + v1m * v1m));
        mom4 += (double)(one_over_n * (v0p * v1p + v0m * v1m));
    }
    return true;
}

// Both coordinates' values at +x and -x from one polynomial each: even part E(z) and odd part O(z), z = x^2,
// p(+x) = E + x O, p(-x) = E - x O.  The constant term goes in last, low part first: one rounding at the
// coordinate's own magnitude, as on the exact path.
// (The coefficients are scalar registers and a gfx9 vector instruction reads only one, so each Horner chain starts with a
// v_mov; keeping the second coefficients in vector registers instead removes 8 instructions per kappa step and was
// SLOWER: 0.342 vs 0.330 ms, measured twice on the same device.)
// (Round 5, the other way round: the two LEADING coefficients of every chain in vector registers for the whole loop, so that a
// chain's first step is fma(vector, vector, scalar) like the others -- 8 instructions and 8 scalar registers less per kappa step,
// no v_mov left in the loop, bit-identical: 0.3141 / 0.3144 -> 0.3278 / 0.3282 ms A/B/A/B on one box, at LOWER socket power and a
// higher clock (1354 W, 2348 MHz against 1373 W, 2298 MHz): the loop waits on its vector-register reads, not on issue slots.)
template <int DEG>
__device__ __forceinline__ void poly_pm(const float* c, float lo_plus, float lo_minus, bool same_lo, float x, float z,
                                        float& plus, float& minus)
{
    static_assert(DEG % 2 == 0 && DEG >= 4 && DEG <= ECC_POLY_DEG, "even degree");
    float E = c[DEG], O = c[DEG - 1];
#pragma unroll
    for (int k = DEG - 2; k >= 2; k -= 2) {
        E = fmaf(E, z, c[k]);
        O = fmaf(O, z, c[k - 1]);
    }
    const float Ep = fmaf(E, z, lo_plus);
    const float Em = same_lo ? Ep : fmaf(E, z, lo_minus);
#ifdef ECC_POLY_UNFUSED_ODD  // rounds 1-3 and most of round 4: the odd part as a rounded product of its own (one instruction more)
    const float xo = x * O;
    plus = (Ep + xo) + c[0];
    minus = (Em - xo) + c[0];
#else
    plus = fmaf(x, O, Ep) + c[0];
    minus = fmaf(-x, O, Em) + c[0];
#endif
}

// Addressing, the one 16-byte load and the bilinear rule for a sample whose coordinates are already known.
// (Non-temporal loads for the view whose band no later pair of the XCD re-uses were measured: they bypass the L1 as
// well and lose the reuse between neighbouring lanes, 0.397 vs 0.338 ms.)
// NOCLAMP: k01's bound on the pair's polynomials (record.poly_ok bit 0) says that neither coordinate can reach a clamp: the
// two v_med3 would return their inputs, and are left out.
template <bool DERIV, int PITCH4, bool NOCLAMP = false>
__device__ __forceinline__ float sample_at(float xa, float yd, unsigned fold, const SlabView sv, float n_t_f, float pitch4_f,
                                           float xa_max)
{
    float fx, fy;
    unsigned off;
    if (PITCH4 > 0) {
        // Cell index and fraction without v_fract_f32 / v_cvt_u32_f32 (quarter rate): x + (2^23 - 0.5) is rounded to the
        // integer 2^23 + rn(x - 0.5) -- the cell floor(x), or x - 1 with fraction 1 when x is an exact integer: the same
        // point of the same bilinear surface -- and leaves the index in the low mantissa bits.  Needs x >= 0.25: the angle
        // coordinate is >= 0.5 in padded texel units; the distance coordinate is clamped at 0.5 instead of 0 (cell 0 is
        // the replicated border: both its bins hold the same bits, the value does not depend on the fraction there).
        if (!NOCLAMP) yd = __builtin_amdgcn_fmed3f(yd, 0.5f, n_t_f);
        // the angle coordinate is within [0.5, n_alpha + 0.5] wherever the fitted polynomial is the mapping it was checked
        // against; between the check points nothing guarantees it, and the index below is 24 mantissa bits times the row
        // pitch: one v_med3 keeps a misbehaving fit inside the slab (advisor, round 3; values in range are not changed)
        if (!NOCLAMP) xa = __builtin_amdgcn_fmed3f(xa, 0.5f, xa_max);
        const float ma = xa + 8388607.5f, md = yd + 8388607.5f;
        fx = xa - (ma - 8388608.f);
        fy = yd - (md - 8388608.f);
        unsigned bin8;  // the low 24 bits of md's pattern (the bin index) times 8: one v_mul_u32_u24 (the compiler would
                        // turn the multiplication by 8 into a shift and a mask)
        asm("v_mul_u32_u24 %0, %1, 8" : "=v"(bin8) : "v"(__float_as_uint(md)));
        off = __umul24(__float_as_uint(ma), (unsigned)PITCH4) + bin8;
    } else {
        yd = __builtin_amdgcn_fmed3f(yd, 0.f, n_t_f);
        xa = __builtin_amdgcn_fmed3f(xa, 0.5f, xa_max);
        fx = __builtin_amdgcn_fractf(xa);
        fy = __builtin_amdgcn_fractf(yd);
        off = footprint_offset<PITCH4>(xa - fx, yd - fy, sv.pitch4, pitch4_f);
    }
    const ecc_v4f_a4 q4 = *(GlobalF4)(sv.origin + off);
    const F4 q = {q4.x, q4.y, q4.z, q4.w};
    const float r0 = fmaf(fx, q.y, q.x);  // q.y, q.w: the row differences, formed when the copy is built
    const float r1 = fmaf(fx, q.w, q.z);
    const float v = fmaf(fy, r1 - r0, r0);
    return DERIV ? __uint_as_float(__float_as_uint(v) ^ fold) : v;
}

// The kappa loop of one pair on the polynomial path (see fit_view_polynomials).
// Measured and dropped: two kappa steps (k, k + 64) per trip with eight gathers in flight per wave -- 0.409 ms against
// 0.400 ms, so the kernel is not short of memory-level parallelism at 8 waves per SIMD; a two-stage software pipeline
// across trips cannot be expressed: the compiler's wait-count insertion drains all loads at the loop header
// (vmcnt(0) before the next stage's loads), and with the gathers in inline assembly it copies their destination
// registers before the explicit wait.
// Later, with the loads and their wait in inline assembly inside ONE trip (gathers of step k issued first, the
// coordinates of step k + 64 computed while they are in flight, explicit s_waitcnt; no loop-carried loaded registers, so
// no early copies -- checked in the ISA, results identical): 0.3305 vs 0.3323 ms, nothing; adding one-byte prefetch
// loads of the next step's four lines (the L2 misses cost 13 %: 0.289 ms with every line cache-resident) made it 0.51 ms,
// two prefetches 0.41 ms -- a gather instruction costs the L1 the same whatever it fetches.
// DEG: the degree the record asks for -- the fit is of degree ECC_POLY_DEG, k01_kernel lowers it where Chebyshev
// economisation costs less than 2e-8 bins (see economise).
// kappa_fit: the end of the polynomials' range (ecc_kappa_fit(kappa_max)); returns the lane's first sample index past it --
// where the exact loop of a pair with kappa_fit < kappa_max carries on.
template <bool DERIV, bool CORR, int PITCH4, int DEG, int WPP = 1, bool NOCLAMP = false>
__device__ __forceinline__ int kappa_loop_poly(int lane, int k_limit, const EccPairRecord* __restrict__ rec, float dkappa,
                                                float kappa_max, float kappa_fit, float w06, const SlabView sv0, const SlabView sv1,
                                                float n_alpha_f, float n_t_f, float pitch4_f, double& acc, double& mom2,
                                                double& mom3, double& mom4, int sub = 0, float* stage = nullptr)
{
    static_assert(WPP == 1 || !CORR, "the shared-pair form carries one term per sample");
    float ca[2][ECC_POLY_DEG + 3], cd[2][ECC_POLY_DEG + 2];
    unsigned fold[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        fold[v] = (unsigned)__builtin_amdgcn_readfirstlane((int)rec->fold[v]);
#pragma unroll
        for (int k = 0; k <= ECC_POLY_DEG + 1; ++k) {
            if (k > DEG && k <= ECC_POLY_DEG) continue;  // dropped coefficients are not even loaded
            ca[v][k] = uniformf(rec->ca[v][k]);
            cd[v][k] = uniformf(rec->cd[v][k]);
        }
        ca[v][ECC_POLY_DEG + 2] = uniformf(rec->ca[v][ECC_POLY_DEG + 2]);
    }
    const float xs = uniformf(rec->x_scale);
    const float xa_max = n_alpha_f + 0.5f;  // padded texel units: row n_alpha of the paired copy is the last one
    // Round 4, -5 of 135 vector instructions per trip, the same bits: (i) the sample index as a float carried along (k < 2^24:
    // exact) instead of a conversion per trip; (ii) the folds' signs.  With the derivative filter a sample is s_v * a_v, s_v = -1
    // where view v's line is folded, both toggled together on the -kappa side; only the RELATIVE sign survives the square:
    // (s0 a0 - s1 a1)^2 = (a0 - s0 s1 a1)^2, and rounding is sign-symmetric -- so the four sign flips per trip become the sign
    // of one wave-uniform factor in the two differences (fma(a1, -+1, a0) is the rounded a0 -+ a1).
    const float rel_sign = (DERIV && ((fold[0] ^ fold[1]) & 0x80000000u)) ? 1.0f : -1.0f;
    const float w06_dkappa = w06 * dkappa;
    float kf = (float)(lane + 64 * sub);
    int k = lane + 64 * sub;
    // Round 5: TWO kappa steps (k, k + 64) per trip while both are inside the range -- eight gathers in flight per wave
    // instead of four.  The launch is bound by the latency of its gathers at the occupancy its scalar registers allow
    // (pairs_kernel.hip, PK_OCCUPANCY): 0.3137 -> 0.3061 ms, 2 986 -> 3 039 evaluations/s A/B/A/B on one box, 72 instead of
    // 45 vector registers (still seven waves per SIMD).  The terms are the single-step loop's, added in its order: same bits.
    // (Round 1 measured the same idea at eight waves per SIMD on the exact path and found nothing.)
    if (WPP == 1 && !CORR) {
        for (; k + 64 < k_limit; k += 128, kf += 128.f) {
            const float kappa_a = dkappa * 0.5f + dkappa * kf;  // ref: ...RadonIntermediate.cu:259 (same fp32 ops)
            const float kappa_b = dkappa * 0.5f + dkappa * (kf + 64.f);
            if (kappa_b >= kappa_fit) break;  // kappa_a < kappa_b: the single-step loop below takes what is left
            const float xA = kappa_a * xs, zA = xA * xA, xB = kappa_b * xs, zB = xB * xB;
            float a0p, a0m, d0p, d0m, a1p, a1m, d1p, d1m, b0p, b0m, e0p, e0m, b1p, b1m, e1p, e1m;
            poly_pm<DEG>(ca[0], ca[0][ECC_POLY_DEG + 1], ca[0][ECC_POLY_DEG + 2], false, xA, zA, a0p, a0m);
            poly_pm<DEG>(cd[0], cd[0][ECC_POLY_DEG + 1], 0.f, true, xA, zA, d0p, d0m);
            poly_pm<DEG>(ca[1], ca[1][ECC_POLY_DEG + 1], ca[1][ECC_POLY_DEG + 2], false, xA, zA, a1p, a1m);
            poly_pm<DEG>(cd[1], cd[1][ECC_POLY_DEG + 1], 0.f, true, xA, zA, d1p, d1m);
            poly_pm<DEG>(ca[0], ca[0][ECC_POLY_DEG + 1], ca[0][ECC_POLY_DEG + 2], false, xB, zB, b0p, b0m);
            poly_pm<DEG>(cd[0], cd[0][ECC_POLY_DEG + 1], 0.f, true, xB, zB, e0p, e0m);
            poly_pm<DEG>(ca[1], ca[1][ECC_POLY_DEG + 1], ca[1][ECC_POLY_DEG + 2], false, xB, zB, b1p, b1m);
            poly_pm<DEG>(cd[1], cd[1][ECC_POLY_DEG + 1], 0.f, true, xB, zB, e1p, e1m);
            const float vA0p = sample_at<false, PITCH4, NOCLAMP>(a0p, d0p, 0u, sv0, n_t_f, pitch4_f, xa_max);
            const float vA1p = sample_at<false, PITCH4, NOCLAMP>(a1p, d1p, 0u, sv1, n_t_f, pitch4_f, xa_max);
            const float vA0m = sample_at<false, PITCH4, NOCLAMP>(a0m, d0m, 0u, sv0, n_t_f, pitch4_f, xa_max);
            const float vA1m = sample_at<false, PITCH4, NOCLAMP>(a1m, d1m, 0u, sv1, n_t_f, pitch4_f, xa_max);
            const float vB0p = sample_at<false, PITCH4, NOCLAMP>(b0p, e0p, 0u, sv0, n_t_f, pitch4_f, xa_max);
            const float vB1p = sample_at<false, PITCH4, NOCLAMP>(b1p, e1p, 0u, sv1, n_t_f, pitch4_f, xa_max);
            const float vB0m = sample_at<false, PITCH4, NOCLAMP>(b0m, e0m, 0u, sv0, n_t_f, pitch4_f, xa_max);
            const float vB1m = sample_at<false, PITCH4, NOCLAMP>(b1m, e1m, 0u, sv1, n_t_f, pitch4_f, xa_max);
            const float pA = fmaf(vA1p, rel_sign, vA0p), mA = fmaf(vA1m, rel_sign, vA0m);
            const float pB = fmaf(vB1p, rel_sign, vB0p), mB = fmaf(vB1m, rel_sign, vB0m);
#ifdef ECC_POLY_UNFUSED_ODD
            acc += (double)(((pA * pA + mA * mA) * w06) * dkappa);
            acc += (double)(((pB * pB + mB * mB) * w06) * dkappa);
#else
            acc += (double)(fmaf(pA, pA, mA * mA) * w06_dkappa);
            acc += (double)(fmaf(pB, pB, mB * mB) * w06_dkappa);
#endif
        }
    }
    for (; k < k_limit; k += 64 * WPP, kf += (float)(64 * WPP)) {
        const float kappa = dkappa * 0.5f + dkappa * kf;  // ref: ...RadonIntermediate.cu:259 (same fp32 ops)
        if (kappa >= kappa_fit) break;
        const float x = kappa * xs, z = x * x;
        float xa0p, xa0m, yd0p, yd0m, xa1p, xa1m, yd1p, yd1m;
        poly_pm<DEG>(ca[0], ca[0][ECC_POLY_DEG + 1], ca[0][ECC_POLY_DEG + 2], false, x, z, xa0p, xa0m);
        poly_pm<DEG>(cd[0], cd[0][ECC_POLY_DEG + 1], 0.f, true, x, z, yd0p, yd0m);
        poly_pm<DEG>(ca[1], ca[1][ECC_POLY_DEG + 1], ca[1][ECC_POLY_DEG + 2], false, x, z, xa1p, xa1m);
        poly_pm<DEG>(cd[1], cd[1][ECC_POLY_DEG + 1], 0.f, true, x, z, yd1p, yd1m);
        // (CORR keeps the signed samples: the cross moment sees the signs)
        const float v0p = sample_at<DERIV && CORR, PITCH4, NOCLAMP>(xa0p, yd0p, fold[0], sv0, n_t_f, pitch4_f, xa_max);
        const float v1p = sample_at<DERIV && CORR, PITCH4, NOCLAMP>(xa1p, yd1p, fold[1], sv1, n_t_f, pitch4_f, xa_max);
        const float v0m = sample_at<DERIV && CORR, PITCH4, NOCLAMP>(xa0m, yd0m, fold[0] ^ 0x80000000u, sv0, n_t_f, pitch4_f, xa_max);
        const float v1m = sample_at<DERIV && CORR, PITCH4, NOCLAMP>(xa1m, yd1m, fold[1] ^ 0x80000000u, sv1, n_t_f, pitch4_f, xa_max);
        if (!CORR) {
            const float vp = fmaf(v1p, rel_sign, v0p), vm = fmaf(v1m, rel_sign, v0m);
#ifdef ECC_POLY_UNFUSED_ODD
            const float consistency = (vp * vp + vm * vm) * w06;  // ref: ...RadonIntermediate.cu:112
            const float term = consistency * dkappa;              // ref: ...RadonIntermediate.cu:269
#else
            // the contracted form of the same two source lines (what nvcc emits for them): one fma, and the two wave-uniform
            // factors as one -- the throughput path only; the per-sample and the reference loops keep every rounding
            const float term = fmaf(vp, vp, vm * vm) * w06_dkappa;
#endif
            if (WPP == 1) acc += (double)term;
            else stage[k] = term;
        } else {
            const float one_over_n = kappa_max / kappa;
            mom2 += (double)(one_over_n * (v0p * v0p + v0m * v0m));
            mom3 += (double)(one_over_n * (v1p * v1p + v1m * v1m));
            mom4 += (double)(one_over_n * (v0p * v1p + v0m * v1m));
        }
    }
    return k;
}


// The kappa loop of one pair, exact per-sample path; k_first: the lane's first sample (lane + 64 sub, or where the polynomial
// loop stopped).
template <bool DERIV, bool CORR, bool REDUCE, int PITCH4, int WPP = 1>
__device__ __forceinline__ void kappa_loop(int k_first, int k_limit, const float (&K0)[8], const float (&K1)[8],
                                           const SlabView sv0, const SlabView sv1, float n_alpha_f, float n_t_f,
                                           float dist_scale, float dist_bias, float pitch4_f, double& acc, double& mom2,
                                           double& mom3, double& mom4, float* stage = nullptr)
{
    const float dkappa = K1[6], kappa_max = K1[7];
    // (Two kappa steps per trip with all eight gathers issued before the first is consumed were measured for the pairs
    // with kappa_max > pi/4, which wait on memory: no change, 0.3349 vs 0.3340 ms for the benchmark's launch.  Again in round 5
    // for the exact part on the row-quad copies, at the kernel's 72-register budget: 36 bytes of scratch and 3 157 / 3 179 / 3 162
    // -> 3 128 / 3 123 / 3 109 evaluations/s A/B/A/B/A/B -- slower.)
    for (int k = k_first; k < k_limit; k += 64 * WPP)
        if (!kappa_step<DERIV, CORR, REDUCE, PITCH4, WPP>(k, K0, K1, dkappa, kappa_max, sv0, sv1, n_alpha_f, n_t_f, dist_scale,
                                                          dist_bias, pitch4_f, acc, mom2, mom3, mom4, stage))
            return;  // kappa only grows: this lane is done
}

// -------------------------------------------------------------------------------------------------
// Sample coordinates as polynomials in kappa.
// For one pair and one of its views the epipolar line of plane angle kappa is l(kappa) = K[:,0] cos + K[:,1] sin,
// and its sample position in the Radon intermediate -- angle coordinate xa(kappa), distance coordinate yd(kappa),
// in padded texel units, after the (alpha+pi, -t) fold -- is an analytic function of kappa on [-kappa_max, kappa_max]
// as long as the fold does not switch.  The reference's "-kappa" sample x = (-cos, sin) is the NEGATED line of
// plane -kappa: the same point of the Radon intermediate with the fold (sign) toggled.  So ONE polynomial pair per
// view serves both signs (evaluated at +x and -x, even and odd parts shared), and it is a very smooth function:
// interpolation of degree 10 at Chebyshev nodes is good to < 1e-6 bins up to kappa_max ~ 1 rad (a C-arm short
// scan: 1e-11 ... 3e-7 bins).  The thread that owns the pair in k01_kernel evaluates the reference's mapping
// (computeK01 lines -> lineToSampleDtr, float constant Pi and float range_t included) at the 11 (symmetric) nodes in float64,
// solves for the monomial coefficients, CHECKS them against the exact mapping at both ends of the range and one
// interior abscissa (tolerance 1e-5 bins, fold state constant) and records the verdict; the pair kernel then spends ~14 instructions per view
// and coordinate on BOTH samples instead of sin/cos, line products, 1/len, a reciprocal, the atan polynomial and
// the fold bookkeeping for each.  Pairs that fail the check (fold inside the range, baseline through the object,
// degenerate geometry) take the exact per-sample path.
// -------------------------------------------------------------------------------------------------
struct CurvePoint {
    double xa, yd;
    bool fold;
    bool valid;
};

// The fit's float64 trigonometry.  No libm and no long series: a table of sin/cos at multiples of pi/64
// (EccPolyTables::sc, correctly rounded on the host) brings every argument within pi/128 of a table angle, where
// five series terms are exact to 1e-19.  (A float64 sincos by its 25-term Taylor series plus a float64 division per
// evaluated point made k01_kernel three times as long.)
#define ECC_TRIG_STEP (3.14159265358979323846 / (2.0 * ECC_TRIG_STEPS))

// The fit's tables as its code sees them: what is indexed per lane (the trig table, the nodes, the check abscissae) staged
// in LDS by the workgroup (a dependent global load costs the fit's chain ~0.6 us each time, and there are five of them
// in a row: theta_ref, a node's sin/cos, its angle, a check's sin/cos, its angle), the inverse Vandermonde matrices --
// indexed by compile-time constants, i.e. scalar loads -- where they are.
struct PolyT {
    const EccPolyTables* g;
    const double (*sc)[2];
    const double* nodes;
    const double* checks;
};

// sin and cos of t in [0, pi/2 + 1e-6]
__device__ __forceinline__ void sincos_table(const PolyT& T, double t, double& s, double& c)
{
    int k = (int)(t * (1.0 / ECC_TRIG_STEP) + 0.5);
    k = min(max(k, 0), ECC_TRIG_STEPS);  // (a NaN converts to 0)
    const double d = fma((double)-k, ECC_TRIG_STEP, t), z = d * d;
    const double sd = d * fma(z, fma(z, fma(z, fma(z, 1.0 / 362880.0, -1.0 / 5040.0), 1.0 / 120.0), -1.0 / 6.0), 1.0);
    const double cd = fma(z, fma(z, fma(z, fma(z, 1.0 / 40320.0, -1.0 / 720.0), 1.0 / 24.0), -0.5), 1.0);
    const double sk = T.sc[k][0], ck = T.sc[k][1];
    s = fma(sk, cd, ck * sd);
    c = fma(ck, cd, -(sk * sd));
}

// Angle of the vector (x, y) in (-pi, pi]: crude float estimate -> rotation back by
// the nearest table angle -> atan of the small remainder by its series; the reciprocal by two Newton steps.
__device__ __forceinline__ double angle_table(const PolyT& T, double x, double y)
{
    const float ax = fabsf((float)x), ay = fabsf((float)y);
    // angle of (|x|, |y|) in units of pi, [0, 1/2], good to 1.3e-3: atan(q) ~ q (pi/4 + 0.273 (1 - q)) on [0, 1] --
    // only the nearest table entry is wanted, the series below absorbs a remainder of pi/128 + 4e-3 (t^11/11 ~ 1e-18)
    const float lo = fminf(ax, ay), hi = fmaxf(ax, ay), q = lo * __builtin_amdgcn_rcpf(hi);
    float a = q * fmaf(-0.0869f, q, 0.3369f);
    if (ay > ax) a = 0.5f - a;
    if (x < 0) a = 1.0f - a;  // angle of (x, |y|) in [0, 1]
    int k = (int)(a * (2.0f * ECC_TRIG_STEPS) + 0.5f);
    k = min(max(k, 0), 2 * ECC_TRIG_STEPS);
    const int kk = k <= ECC_TRIG_STEPS ? k : 2 * ECC_TRIG_STEPS - k;
    const double sk = T.sc[kk][0], ck = k <= ECC_TRIG_STEPS ? T.sc[kk][1] : -T.sc[kk][1];
    const double yy = fabs(y);
    const double u = fma(x, ck, yy * sk), w = fma(yy, ck, -(x * sk));  // (x, |y|) rotated by -k pi/64: u = length * cos(rest) > 0
    double r = (double)__builtin_amdgcn_rcpf((float)u);
    r = r * fma(-u, r, 2.0);
    r = r * fma(-u, r, 2.0);
    const double t = w * r, z = t * t;
    const double d = t * fma(z, fma(z, fma(z, fma(z, 1.0 / 9.0, -1.0 / 7.0), 1.0 / 5.0), -1.0 / 3.0), 1.0);
    const double ang = fma((double)k, ECC_TRIG_STEP, d);
    return y < 0 ? -ang : ang;
}

// The reference's mapping line -> (angle, distance) texel coordinates for one point of a view's curve
// (ref: getRedundancy's line, ...RadonIntermediate.cu:74; lineToSampleDtr, EpipolarConsistencyCommon.hxx:152-171;
// the texel mapping of a normalised texture), in float64 with the reference's float constants, one coordinate
// at a time.
// The angle of (l0, l1) is taken RELATIVE to the curve's line at kappa = 0, whose normal (K[0], K[1]) is a unit
// vector by construction (shiftOriginAndNormlaize): theta = theta_ref + D, D = angle of (dot, cross) by angle_table.
// 1/len: float rsqrt + two Newton steps.  Not valid (-> the pair takes the exact path) when the line turns by 90
// degrees or more against kappa = 0.  The fold state is decided by the angle.
struct CurveGeom {
    double theta_ref, inv_range_t, n_alpha, n_t;
    // l(kappa) = A cos + B sin with A = (K[0], K[1]), B = (K[3], K[4]): nn = A.A, alpha = A.B, beta = A x B, gamma = B.B
    double nn, alpha, beta, gamma, k2, k5;
};

__device__ __forceinline__ void curve_geometry(const float* K, CurveGeom& g)
{
    const double a0 = K[0], a1 = K[1], b0 = K[3], b1 = K[4];
    g.nn = fma(a0, a0, a1 * a1);
    g.alpha = fma(a0, b0, a1 * b1);
    g.beta = fma(a0, b1, -(a1 * b0));
    g.gamma = fma(b0, b0, b1 * b1);
    g.k2 = K[2];
    g.k5 = K[5];
}

__device__ __forceinline__ double exact_angle_coord(const PolyT& T, const CurveGeom& g, double c, double s, bool& fold,
                                                    bool& valid)
{
    const double pi = 3.14159265358979323846, inv_Pi_f = 1.0 / (double)3.14159265359f;
    const double dot = fma(g.nn, c, g.alpha * s);  // l . A
    const double cross = g.beta * s;               // A x l
    valid = dot > 0.0 && fabs(cross) < 1e30 && dot < 1e30;
    const double D = angle_table(T, dot, cross);
    double theta = g.theta_ref + D;     // in (-3pi/2, 3pi/2): bring back to atan2's range (-pi, pi]
    if (theta > pi) theta -= 2.0 * pi;
    else if (theta <= -pi) theta += 2.0 * pi;
    double a = theta * inv_Pi_f;
    if (a < 0) a += 2;
    fold = a > 1;
    if (fold) a -= 1;
    return fma(a, g.n_alpha, 0.5);  // texel position a*n_alpha - .5, +1 for the slab's border row
}

__device__ __forceinline__ double exact_distance_coord(const CurveGeom& g, double c, double s, bool fold)
{
    const double as = g.alpha * s;
    const double len2 = fma(c, fma(g.nn, c, as + as), g.gamma * s * s);  // |A c + B s|^2
    const double l2 = fma(g.k2, c, g.k5 * s), h = -0.5 * len2;
    double y = (double)__builtin_amdgcn_rsqf((float)len2);
    y = y * fma(h * y, y, 1.5);
    y = y * fma(h * y, y, 1.5);  // relative error ~1e-7 -> 1e-14 -> 1e-28
    double d = fma(-(l2 * y), g.inv_range_t, 0.5);
    if (fold) d = 1 - d;
    return fma(d, g.n_t, 0.5);
}

// Fits ONE coordinate (ANGLE: the angle coordinate xa, else the distance coordinate yd) of one view of one pair and
// checks it; c: DEG+1 float64 monomial coefficients in x = kappa / kappa_max.  The distance coordinate is fitted in
// the direct (unfolded) state: the fold is the reflection yd -> n_t + 1 - yd, applied by the caller once the thread
// that fits the angle has decided it.
// The coefficients are accumulated node by node (c_2k += Ae[k][j] fe_j, c_2k+1 += Ao[k][j] fo_j with the even / odd
// combinations fe, fo of the values at +-x_j); the node loop stays rolled: unrolled, the kernel spilled 150 scalar
// registers and spent a fifth of its instructions moving constants.
template <bool ANGLE>
__device__ bool fit_coordinate(const PolyT& T, const CurveGeom& g, double km, bool& fold0, double* c)
{
    constexpr int N = ECC_POLY_DEG + 1, H = ECC_POLY_DEG / 2;
    bool ok = true;
    fold0 = false;
#pragma unroll
    for (int k = 0; k < N; ++k) c[k] = 0.0;
#pragma unroll 1
    for (int j = 0; j < H; ++j) {
        const double xj = T.nodes[j];
        double sn, cs, fp, fm;
        sincos_table(T, xj * km, sn, cs);
        if (ANGLE) {
            bool f1 = false, f2 = false, v1 = true, v2 = true;
            fp = exact_angle_coord(T, g, cs, sn, f1, v1);
            if (j == 0) fold0 = f1;
            fm = exact_angle_coord(T, g, cs, -sn, f2, v2);  // the mirrored node: cos(-t) = cos t, sin(-t) = -sin t
            ok = ok && v1 && v2 && f1 == fold0 && f2 == fold0;
        } else {
            fp = exact_distance_coord(g, cs, sn, false);
            fm = exact_distance_coord(g, cs, -sn, false);
        }
        const double fe = 0.5 * (fp + fm), fo = (fp - fm) * (0.5 / xj);
#pragma unroll
        for (int k = 0; k <= H; ++k) c[2 * k] = fma(T.g->Ae[k * (H + 1) + j], fe, c[2 * k]);
#pragma unroll
        for (int k = 0; k < H; ++k) c[2 * k + 1] = fma(T.g->Ao[k * H + j], fo, c[2 * k + 1]);
    }
    {   // the centre node, kappa = 0
        bool f1 = false, v1 = true;
        const double f0 = ANGLE ? exact_angle_coord(T, g, 1.0, 0.0, f1, v1) : exact_distance_coord(g, 1.0, 0.0, false);
        if (ANGLE) ok = ok && v1 && f1 == fold0;
#pragma unroll
        for (int k = 0; k <= H; ++k) c[2 * k] = fma(T.g->Ae[k * (H + 1) + H], f0, c[2 * k]);
    }
    // the check measures the interpolation error (float64 coefficients); the float rounding of the
    // coefficients is evaluation noise of the same kind as the exact path's own fp32 rounding
#pragma unroll 1
    for (int j = 0; j < ECC_POLY_CHECKS; ++j) {
        const double x = T.checks[j];
        double sn, cs;
        sincos_table(T, fabs(x) * km, sn, cs);
        if (x < 0) sn = -sn;
        bool f = fold0, v = true;
        const double q = ANGLE ? exact_angle_coord(T, g, cs, sn, f, v) : exact_distance_coord(g, cs, sn, false);
        double pq = c[N - 1];
#pragma unroll
        for (int k = N - 2; k >= 0; --k) pq = fma(pq, x, c[k]);
        ok = ok && v && f == fold0 && fabs(pq - q) <= 1e-5;  // NaN fails
    }
    return ok;
}

// The same fit with the nodes and the checks spread over the LANES = 8 adjacent lanes a pair has in k01_kernel<8> (small
// launches: the kernel's time is the length of one thread's chain of dependent float64 operations).  Lane j < H evaluates
// node j (both signs), lane H the centre node; the even / odd node values are then exchanged inside the group and EVERY
// lane accumulates the coefficients in the order fit_coordinate does (j = 0 .. H-1, then the centre), so c -- and with it
// every pair value -- is bit-identical to the one-thread fit; check q runs on lane q.  The verdict is the AND over the group.
template <bool ANGLE, int LANES>
__device__ bool fit_coordinate_wide(const PolyT& T, const CurveGeom& g, double km, bool& fold0, double* c, int j)
{
    constexpr int N = ECC_POLY_DEG + 1, H = ECC_POLY_DEG / 2;
    static_assert(LANES >= H + 1 && LANES >= ECC_POLY_CHECKS && (LANES & (LANES - 1)) == 0, "one lane per node and per check");
    const int lane = threadIdx.x & 63, base = lane & ~(LANES - 1);
    if constexpr (LANES >= 2 * H + 1 + ECC_POLY_CHECKS) {
        // 16 lanes per fit: ONE exact curve point per lane -- lanes 0 .. H-1 the nodes at +kappa, H .. 2H-1 the same nodes at
        // -kappa, 2H the centre, the next ECC_POLY_CHECKS the check points, all at once; then the same exchanges and the same
        // accumulation order as below.  The chain of a fit is one curve point + the accumulation instead of three points.
        constexpr int C0 = 2 * H + 1;
        const bool neg = j >= H && j < 2 * H, is_check = j >= C0 && j < C0 + ECC_POLY_CHECKS;
        bool ok16 = true, f = false, v = true;
        double val = 0.0, xq = 0.0;
        if (j < C0 + ECC_POLY_CHECKS) {
            double sn = 0.0, cs = 1.0;
            if (j < 2 * H) {
                sincos_table(T, T.nodes[neg ? j - H : j] * km, sn, cs);
                if (neg) sn = -sn;  // the mirrored node: cos(-t) = cos t, sin(-t) = -sin t
            } else if (is_check) {
                xq = T.checks[j - C0];
                sincos_table(T, fabs(xq) * km, sn, cs);
                if (xq < 0) sn = -sn;
            }
            val = ANGLE ? exact_angle_coord(T, g, cs, sn, f, v) : exact_distance_coord(g, cs, sn, false);
            ok16 = v;
        }
        fold0 = __shfl((int)f, base) != 0;  // the state at node 0, +kappa (what fit_coordinate takes)
        if (ANGLE && j < C0 + ECC_POLY_CHECKS) ok16 = ok16 && f == fold0;
#pragma unroll
        for (int k = 0; k < N; ++k) c[k] = 0.0;
#pragma unroll
        for (int jj = 0; jj < H; ++jj) {
            const double fp = __shfl(val, base + jj), fm = __shfl(val, base + H + jj);
            const double fej = 0.5 * (fp + fm), foj = (fp - fm) * (0.5 / T.nodes[jj]);
#pragma unroll
            for (int k = 0; k <= H; ++k) c[2 * k] = fma(T.g->Ae[k * (H + 1) + jj], fej, c[2 * k]);
#pragma unroll
            for (int k = 0; k < H; ++k) c[2 * k + 1] = fma(T.g->Ao[k * H + jj], foj, c[2 * k + 1]);
        }
        {
            const double f0 = __shfl(val, base + 2 * H);
#pragma unroll
            for (int k = 0; k <= H; ++k) c[2 * k] = fma(T.g->Ae[k * (H + 1) + H], f0, c[2 * k]);
        }
        if (is_check) {
            double pq = c[N - 1];
#pragma unroll
            for (int k = N - 2; k >= 0; --k) pq = fma(pq, xq, c[k]);
            ok16 = ok16 && fabs(pq - val) <= 1e-5;  // NaN fails
        }
        const unsigned long long all16 = __ballot(ok16);
        return ((all16 >> base) & ((1ull << LANES) - 1)) == ((1ull << LANES) - 1);
    }
    bool ok = true, f1 = false;
    double fe = 0.0, fo = 0.0;
    if (j < H) {
        const double xj = T.nodes[j];
        double sn, cs;
        sincos_table(T, xj * km, sn, cs);
        double fp, fm;
        if (ANGLE) {
            bool f2 = false, v1 = true, v2 = true;
            fp = exact_angle_coord(T, g, cs, sn, f1, v1);
            fm = exact_angle_coord(T, g, cs, -sn, f2, v2);
            ok = v1 && v2 && f1 == f2;  // both against node 0's state below
        } else {
            fp = exact_distance_coord(g, cs, sn, false);
            fm = exact_distance_coord(g, cs, -sn, false);
        }
        fe = 0.5 * (fp + fm);
        fo = (fp - fm) * (0.5 / xj);
    } else if (j == H) {
        bool v1 = true;
        fe = ANGLE ? exact_angle_coord(T, g, 1.0, 0.0, f1, v1) : exact_distance_coord(g, 1.0, 0.0, false);
        ok = v1;
    }
    fold0 = __shfl((int)f1, base) != 0;  // the state at node 0, +kappa (what fit_coordinate takes)
    if (ANGLE && j <= H) ok = ok && f1 == fold0;
#pragma unroll
    for (int k = 0; k < N; ++k) c[k] = 0.0;
#pragma unroll
    for (int jj = 0; jj < H; ++jj) {
        const double fej = __shfl(fe, base + jj), foj = __shfl(fo, base + jj);
#pragma unroll
        for (int k = 0; k <= H; ++k) c[2 * k] = fma(T.g->Ae[k * (H + 1) + jj], fej, c[2 * k]);
#pragma unroll
        for (int k = 0; k < H; ++k) c[2 * k + 1] = fma(T.g->Ao[k * H + jj], foj, c[2 * k + 1]);
    }
    {
        const double f0 = __shfl(fe, base + H);
#pragma unroll
        for (int k = 0; k <= H; ++k) c[2 * k] = fma(T.g->Ae[k * (H + 1) + H], f0, c[2 * k]);
    }
    if (j < ECC_POLY_CHECKS) {
        const double x = T.checks[j];
        double sn, cs;
        sincos_table(T, fabs(x) * km, sn, cs);
        if (x < 0) sn = -sn;
        bool f = fold0, v = true;
        const double q = ANGLE ? exact_angle_coord(T, g, cs, sn, f, v) : exact_distance_coord(g, cs, sn, false);
        double pq = c[N - 1];
#pragma unroll
        for (int k = N - 2; k >= 0; --k) pq = fma(pq, x, c[k]);
        ok = ok && v && f == fold0 && fabs(pq - q) <= 1e-5;  // NaN fails
    }
    const unsigned long long all = __ballot(ok);
    return ((all >> base) & ((1ull << LANES) - 1)) == ((1ull << LANES) - 1);
}

// Chebyshev economisation of the fitted polynomial: x^n = (T_n(x) + lower powers) / 2^(n-1) on [-1, 1], so the two top
// monomials can be folded into the lower ones at an error of at most |c_n| / 2^(n-1) + |c_(n-1)| / 2^(n-2) -- two to
// three orders of magnitude less than dropping them.  Lowers the degree two at a time while the accumulated bound
// stays below 2e-8 bins and returns the degree the pair kernel has to evaluate (10, 8, 6 or 4); c is updated.
__device__ __forceinline__ int economise(double* c, double tol)
{
    static_assert(ECC_POLY_DEG == 10, "degree classes 4 / 6 / 8 / 10");
    double err = fabs(c[10]) * (1.0 / 512.0) + fabs(c[9]) * (1.0 / 256.0);
    if (!(err <= tol)) return 10;
    {   // T10 = 512x^10 - 1280x^8 + 1120x^6 - 400x^4 + 50x^2 - 1,  T9 = 256x^9 - 576x^7 + 432x^5 - 120x^3 + 9x
        const double a = c[10] * (1.0 / 512.0), b = c[9] * (1.0 / 256.0);
        c[8] = fma(a, 1280.0, c[8]); c[6] = fma(a, -1120.0, c[6]); c[4] = fma(a, 400.0, c[4]); c[2] = fma(a, -50.0, c[2]); c[0] += a;
        c[7] = fma(b, 576.0, c[7]); c[5] = fma(b, -432.0, c[5]); c[3] = fma(b, 120.0, c[3]); c[1] = fma(b, -9.0, c[1]);
        c[10] = c[9] = 0.0;
    }
    err += fabs(c[8]) * (1.0 / 128.0) + fabs(c[7]) * (1.0 / 64.0);
    if (!(err <= tol)) return 8;
    {   // T8 = 128x^8 - 256x^6 + 160x^4 - 32x^2 + 1,  T7 = 64x^7 - 112x^5 + 56x^3 - 7x
        const double a = c[8] * (1.0 / 128.0), b = c[7] * (1.0 / 64.0);
        c[6] = fma(a, 256.0, c[6]); c[4] = fma(a, -160.0, c[4]); c[2] = fma(a, 32.0, c[2]); c[0] -= a;
        c[5] = fma(b, 112.0, c[5]); c[3] = fma(b, -56.0, c[3]); c[1] = fma(b, 7.0, c[1]);
        c[8] = c[7] = 0.0;
    }
    err += fabs(c[6]) * (1.0 / 32.0) + fabs(c[5]) * (1.0 / 16.0);
    if (!(err <= tol)) return 6;
    {   // T6 = 32x^6 - 48x^4 + 18x^2 - 1,  T5 = 16x^5 - 20x^3 + 5x
        const double a = c[6] * (1.0 / 32.0), b = c[5] * (1.0 / 16.0);
        c[4] = fma(a, 48.0, c[4]); c[2] = fma(a, -18.0, c[2]); c[0] += a;
        c[3] = fma(b, 20.0, c[3]); c[1] = fma(b, -5.0, c[1]);
        c[6] = c[5] = 0.0;
    }
    return 4;
}

template <int LANES>
struct K01Shared {
    EccPairRecord recs[64 / LANES];
    int ok_flags[4][64 / LANES];
    // the per-lane-indexed tables of the fit (PolyT)
    double sc[ECC_TRIG_STEPS + 1][2];
    double nodes[ECC_POLY_DEG + 1];
    double checks[ECC_POLY_CHECKS];
};

// The records of the pairs blk_first + slot, slot < min(live_slots, 64 / LANES), of the launch p, assembled in sh.recs
// (LDS).  Called by all threads of a workgroup; the first 256 are its members (member: wave-uniform; workgroups of more
// threads pass false for the others, which only take part in the barriers); ends with a barrier (the records are complete
// for everybody).
// small (small_eval_kernel.hip): the index tuples of the workgroup's pairs come from idx_lds (4 ints per slot, read from
// the caller's list once per workgroup) and the geometry of the few views whose matrix changed from the kernel arguments
// (small->patch_*); workgroup 0 copies those entries into the device arrays, which no thread reads for a patched view.
// (`small` points INTO THE KERNEL-ARGUMENT SEGMENT -- constant address space, scalar loads -- not at a by-value copy of the
// argument: indexing such a copy with a run-time index makes the compiler move all of it to scratch memory.)
// coefficient of T_k in x^n (n >= k, same parity): 2^(1-n) C(n, (n-k)/2), halved for k = 0 -- a compile-time constant wherever
// it is used (the loops over n and k are unrolled)
__device__ __forceinline__ constexpr double cheb_weight(int n, int k)
{
    double binom = 1.0;
    const int r = (n - k) / 2;
    for (int i = 1; i <= r; ++i) binom = binom * (double)(n - r + i) / (double)i;
    double w = binom;
    for (int i = 1; i < n; ++i) w *= 0.5;  // 2^(1-n) for n >= 1
    if (n == 0) w = 1.0;
    return k == 0 && n > 0 ? 0.5 * w : w;
}

typedef const EccSmallEval __attribute__((address_space(4))) * EccSmallEvalArg;
template <int LANES>
__device__ __forceinline__ void k01_fit_block(const EccPairParams& p, long long blk_first, int live_slots, K01Shared<LANES>& sh,
                                              EccSmallEvalArg small = nullptr, const int32_t* idx_lds = nullptr, const bool member = true)
{
    constexpr int N = ECC_POLY_DEG + 1;
    const int role = (threadIdx.x >> 6) & 3, v = role & 1, slot = (threadIdx.x & 63) / LANES, jl = threadIdx.x & (LANES - 1);
    const bool angle_role = role < 2;
    const long long local = blk_first + slot;
    const bool live = member && slot < live_slots && local < p.count;
    if (p.poly && member) {  // p.poly: uniform over the launch: stage the tables (complete at the barrier in front of the fit)
        constexpr int N_SC = 2 * (ECC_TRIG_STEPS + 1);
        const int q = threadIdx.x;
        if (q < N_SC) (&sh.sc[0][0])[q] = (&p.poly->sc[0][0])[q];
        else if (q < N_SC + ECC_POLY_DEG + 1) sh.nodes[q - N_SC] = p.poly->nodes[q - N_SC];
        else if (q < N_SC + ECC_POLY_DEG + 1 + ECC_POLY_CHECKS) sh.checks[q - N_SC - ECC_POLY_DEG - 1] = p.poly->checks[q - N_SC - ECC_POLY_DEG - 1];
    }
    int iP0 = 0, iP1 = 0, iD0 = 0, iD1 = 0, ci = 0, cj = 0;
    if (live) {
        if (idx_lds) {
            const int32_t* q = idx_lds + 4 * slot;
            iP0 = q[0]; iP1 = q[1]; iD0 = q[2]; iD1 = q[3];
        } else if (p.indices) {
            const int32_t* q = p.indices + 4 * (p.first + local);
            iP0 = q[0]; iP1 = q[1]; iD0 = q[2]; iD1 = q[3];
        } else {
            ecc_get_ij_device(p.first + local, p.n_views, ci, cj);
            iP0 = iD0 = ci;
            iP1 = iD1 = cj;
        }
    }
    // the thread's view's half of K01: Kv[0..5] the line map, Kv[6..7] = (baseline distance, view angle) for view 0,
    // (dkappa, kappa_max) for view 1 -- the layout of the reference's K01 array
    float Kv[8], kappa_max = 0.f, dkappa = 0.f;
    for (int i = 0; i < 8; i++) Kv[i] = 0.f;
    // Record reuse: views whose matrix changed since the kept records were made come from the host's patch list (the
    // same E1 arithmetic, ecc_host_geometry.h); workgroup 0 also copies the list into the device arrays, which no
    // thread of this launch reads for a patched view.
    if (p.patch_count > 0 && blockIdx.x == 0 && member)
        for (int q = threadIdx.x; q < 16 * p.patch_count; q += 256) {
            const int e = q >> 4, w = q & 15, view = p.patch_views[e];
            const float val = p.patch_geo[q];
            if (w < 12) const_cast<float*>(p.PinvTs)[12 * view + w] = val;
            else const_cast<float*>(p.Cs)[4 * view + (w - 12)] = val;
        }
    if (small && small->patch_count > 0 && blockIdx.x == 0 && member)
        for (int q = threadIdx.x; q < 16 * small->patch_count; q += 256) {
            const int e = q >> 4, w = q & 15, view = small->patch_views[e];
            const float val = small->patch_geo[e][w];
            if (w < 12) const_cast<float*>(p.PinvTs)[12 * view + w] = val;
            else const_cast<float*>(p.Cs)[4 * view + (w - 12)] = val;
        }
    if (live && p.record_slots) {  // the kept records serve all-pairs launches: cost-image position = the pair itself
        ci = iP0;
        cj = iP1;
    }
    if (live && iP0 != iP1) {
        float Kp[8], s2, s3, K06, K07;
        const float *C0 = p.Cs + 4 * iP0, *C1 = p.Cs + 4 * iP1, *Pv = p.PinvTs + 12 * (v ? iP1 : iP0);
        if (small) {
            int e0 = -1, e1 = -1;  // entries of the patch list that hold view iP0 / iP1 (an index, then ONE load each)
#pragma unroll 1
            for (int e = 0; e < small->patch_count; ++e) {  // at most ECC_SMALL_PATCH_MAX entries, kernel arguments
                const int view = small->patch_views[e];
                e0 = view == iP0 ? e : e0;
                e1 = view == iP1 ? e : e1;
            }
            if (e0 >= 0) C0 = (const float*)&small->patch_geo[e0][12];
            if (e1 >= 0) C1 = (const float*)&small->patch_geo[e1][12];
            const int ev = v ? e1 : e0;
            if (ev >= 0) Pv = (const float*)&small->patch_geo[ev][0];
        } else if (p.patch_ref) {
            const int r0 = p.patch_ref[2 * (p.first + local)], r1 = p.patch_ref[2 * (p.first + local) + 1];
            if (r0 >= 0) C0 = p.patch_geo + 16 * r0 + 12;
            if (r1 >= 0) C1 = p.patch_geo + 16 * r1 + 12;
            const int rv = v ? r1 : r0;
            if (rv >= 0) Pv = p.patch_geo + 16 * rv;
        }
        baseline_pencil(C0, C1, Kp, s2, s3);
        project_pencil(Pv, Kp, p.n_x2, p.n_y2, Kv);
        pencil_range(s2, s3, p.object_radius_mm, p.num_samples, p.dkappa_user, p.K01_out != nullptr && angle_role, K06, K07,
                     dkappa, kappa_max);
        Kv[6] = v ? dkappa : K06;
        Kv[7] = v ? kappa_max : K07;
    }

    static_assert(sizeof(EccPairRecord) % 8 == 0, "record copied as 8-byte words");
    EccPairRecord* r = &sh.recs[slot];

    double c[N];
    bool fold0 = false;
    int ok = 0;
    if (p.poly) __syncthreads();  // the staged tables are complete (uniform over the launch)
    const PolyT T = {p.poly, sh.sc, sh.nodes, sh.checks};
    if (live && p.poly && kappa_max > 0.f && dkappa > 0.f) {
        CurveGeom g;
        g.theta_ref = angle_role ? angle_table(T, (double)Kv[0], (double)Kv[1]) : 0.0;
        g.inv_range_t = 1.0 / (double)p.range_t;
        g.n_alpha = (double)p.n_alpha;
        g.n_t = (double)p.n_t;
        curve_geometry(Kv, g);
        if (LANES == 1) {
            if (angle_role) ok = fit_coordinate<true>(T, g, (double)ecc_kappa_fit(kappa_max), fold0, c) ? 1 : 0;  // wave-uniform branch
            else ok = fit_coordinate<false>(T, g, (double)ecc_kappa_fit(kappa_max), fold0, c) ? 1 : 0;
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) c[k] = 0.0;
    }
    if constexpr (LANES > 1) if (p.poly && member) {  // p.poly: uniform over the launch
        // every lane takes part in the exchanges of the wide fit; a group without a fit (dead slot, empty kappa range --
        // the same for all its lanes) discards the result
        const bool fit = live && kappa_max > 0.f && dkappa > 0.f;
        CurveGeom g;
        g.theta_ref = (fit && angle_role) ? angle_table(T, (double)Kv[0], (double)Kv[1]) : 0.0;
        g.inv_range_t = 1.0 / (double)p.range_t;
        g.n_alpha = (double)p.n_alpha;
        g.n_t = (double)p.n_t;
        curve_geometry(Kv, g);
        double cw[N];
        bool fw = false;
        const bool okw = angle_role ? fit_coordinate_wide<true, LANES>(T, g, fit ? (double)ecc_kappa_fit(kappa_max) : 0.0, fw, cw, jl)
                                    : fit_coordinate_wide<false, LANES>(T, g, fit ? (double)ecc_kappa_fit(kappa_max) : 0.0, fw, cw, jl);
        if (fit) {
            ok = okw ? 1 : 0;
            fold0 = fw;
#pragma unroll
            for (int k = 0; k < N; ++k) c[k] = cw[k];
        }
    }
    if (ok) ok = economise(c, (double)p.economise_tol);
    // Can a sample of this coordinate reach a clamp of the pair kernel (angle: [0.5, n_alpha + 0.5], distance: [0.5, n_t])?
    // |p(x) - a0| <= sum |a_k| on [-1, 1] for the Chebyshev coefficients a_k of p (both signs of kappa; the distance bound
    // also holds for the reflected polynomial n_t + 1 - p, which is what is stored when the view is folded); 0.05 bins cover
    // the float evaluation and the low parts.
    int in_range = 0;
    if (ok) {
        // in the Chebyshev basis (|T_k| <= 1 on [-1, 1]; the monomial sum over-estimates the range of a wide curve several
        // times): x^n = 2^(1-n) sum_k C(n, (n-k)/2) T_k over k = n, n-2, ..., the k = 0 term halved
        double a0 = 0.0, spread = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            double ak = 0.0;
#pragma unroll
            for (int nn = k; nn < N; nn += 2) ak = fma(c[nn], cheb_weight(nn, k), ak);
            if (k == 0) a0 = ak;
            else spread += fabs(ak);
        }
        const double lo = a0 - spread, hi = a0 + spread, margin = 0.05;
        in_range = angle_role ? (lo >= 0.5 + margin && hi <= (double)p.n_alpha + 0.5 - margin)
                              : (lo >= 1.0 + margin && hi <= (double)p.n_t - margin);
    }
    const bool writer = member && jl == 0;  // LANES > 1: the lanes of a group hold identical results
    if (writer) sh.ok_flags[role][slot] = ok | in_range;  // degree (even) | 1
    if (angle_role && writer) {
        r->fold[v] = fold0 ? 0x80000000u : 0u;
#pragma unroll
        for (int k = 0; k < N; ++k) r->ca[v][k] = (float)c[k];
        const float c0 = (float)c[0];
        r->ca[v][N] = (float)(c[0] - (double)c0);  // low part of the constant term
        // The negated line of the -kappa sample is in the OTHER fold state.  With the reference's float Pi
        // (= pi (1 + 2.78e-8), EpipolarConsistencyCommon.hxx:155,159) the direct branch gives a = r (1 - e) and the
        // folded one a = r (1 - e) + e for the same geometric line, e = 1 - pi / Pi: a constant offset of
        // e * n_alpha bins between the two states (2.1e-5 bins at 768 -- the systematic shift DESIGN.md 2 is about).
        const double Pi_f = (double)3.14159265359f, e = 1.0 - 3.14159265358979323846 / Pi_f;
        const double delta = (fold0 ? -e : e) * (double)p.n_alpha;
        r->ca[v][N + 1] = (float)(c[0] + delta - (double)c0);
        float* Kdst = v ? r->K1 : r->K0;
        for (int i = 0; i < 8; i++) Kdst[i] = Kv[i];
        if (live && p.K01_out)
            for (int i = 0; i < 8; i++) p.K01_out[16 * local + 8 * v + i] = Kv[i];
    }
    __syncthreads();
    if (!angle_role && writer) {
        if (r->fold[v]) {  // this view's fold, decided by the angle's thread: yd -> n_t + 1 - yd
            c[0] = (double)p.n_t + 1.0 - c[0];
#pragma unroll
            for (int k = 1; k < N; ++k) c[k] = -c[k];
        }
#pragma unroll
        for (int k = 0; k < N; ++k) r->cd[v][k] = (float)c[k];
        r->cd[v][N] = (float)(c[0] - (double)(float)c[0]);
    }
    if (role == 0 && writer) {
        r->iD0 = iD0;
        r->iD1 = iD1;
        r->ci = ci;
        r->cj = cj;
        const int d0 = sh.ok_flags[0][slot], d1 = sh.ok_flags[1][slot], d2 = sh.ok_flags[2][slot], d3 = sh.ok_flags[3][slot];
        const int degree = (d0 && d1 && d2 && d3) ? max(max(d0 & ~1, d1 & ~1), max(d2 & ~1, d3 & ~1)) : 0;
        r->poly_ok = degree ? (degree | (d0 & d1 & d2 & d3 & 1)) : 0;  // bit 0: no sample can reach a clamp
        r->x_scale = degree ? (float)(1.0 / (double)ecc_kappa_fit(kappa_max)) : 0.f;  // the polynomials' range (ecc_layout.h)
    }
    __syncthreads();
}


// The sampling loops of ONE pair in one wave (ref: kernelEpipolarCosistency, ...RadonIntermediate.cu:214-276): the record's
// polynomials where k01's fit was accepted, the exact per-sample path otherwise; per-lane float64 partial sums in acc
// (CORR: the three moments).  iD0 / iD1: the pair's Radon intermediates (already read from the record by the caller).
// WPP > 1: this wave takes the trips sub, sub + WPP, ... of the pair and stores each sample's term into stage[k].
template <bool DERIV, bool CORR, int WPP = 1>
__device__ __forceinline__ void pair_accumulate(const EccPairParams& p, const EccPairRecord* __restrict__ rec, int iD0, int iD1,
                                                int lane, double& acc, double& mom2, double& mom3, double& mom4, int sub = 0,
                                                float* stage = nullptr)
{
    const unsigned pitch4 = (unsigned)p.pitch * 8u;  // row pitch of the paired copies in bytes
    const SlabView sv0 = {(GlobalBytes)p.dtrs[iD0], pitch4};
    const SlabView sv1 = {(GlobalBytes)p.dtrs[iD1], pitch4};
    const float n_alpha_f = (float)p.n_alpha, n_t_f = (float)p.n_t;
    const float pitch4_f = (float)pitch4;
    const float kappa_max = uniformf(rec->K1[7]);

    const bool reduce = kappa_max > 0.785398163397448f;  // wave-uniform
    const int poly_raw = __builtin_amdgcn_readfirstlane(rec->poly_ok);
    const int poly_ok = poly_raw & ~1;       // the degree
    const bool in_range = (poly_raw & 1) != 0;  // no sample of this pair can reach a clamp (k01_fit_block's bound)
    int k_first = lane + 64 * sub;  // the lane's first sample of the exact loop
    if (poly_ok) {
        const float kappa_fit = ecc_kappa_fit(kappa_max), dkappa = uniformf(rec->K1[6]), w06 = uniformf(rec->K0[6]);
#define ECC_POLY_LOOP_NC(P4, DEG, NC) \
    k_first = kappa_loop_poly<DERIV, CORR, P4, DEG, WPP, NC>(lane, p.k_limit, rec, dkappa, kappa_max, kappa_fit, w06, sv0, sv1, n_alpha_f, n_t_f, pitch4_f, acc, mom2, mom3, mom4, sub, stage)
#define ECC_POLY_LOOP(P4, DEG) ECC_POLY_LOOP_NC(P4, DEG, false)
        if (p.wide_offsets) {
            if (poly_ok <= 6) ECC_POLY_LOOP(-1, 6);
            else ECC_POLY_LOOP(-1, ECC_POLY_DEG);
        } else if (pitch4 == 6400u && in_range) {  // (the clamp-free loops exist for the default 768 distance bins only)
            if (poly_ok <= 4) ECC_POLY_LOOP_NC(6400, 4, true);
            else if (poly_ok <= 6) ECC_POLY_LOOP_NC(6400, 6, true);
            else if (poly_ok <= 8) ECC_POLY_LOOP_NC(6400, 8, true);
            else ECC_POLY_LOOP_NC(6400, ECC_POLY_DEG, true);
        } else if (pitch4 == 6400u) {
            if (poly_ok <= 4) ECC_POLY_LOOP(6400, 4);
            else if (poly_ok <= 6) ECC_POLY_LOOP(6400, 6);
            else if (poly_ok <= 8) ECC_POLY_LOOP(6400, 8);
            else ECC_POLY_LOOP(6400, ECC_POLY_DEG);
        } else {
            if (poly_ok <= 6) ECC_POLY_LOOP(0, 6);
            else ECC_POLY_LOOP(0, ECC_POLY_DEG);
        }
#undef ECC_POLY_LOOP
#undef ECC_POLY_LOOP_NC
        if (!(kappa_fit < kappa_max)) return;  // wave-uniform: the polynomials covered the whole range (the normal case)
        // what follows is read from the record afterwards: nothing of the exact loop occupies a register during the loops above
        // (their scalar registers are the kernel's occupancy limit, pairs_kernel.hip)
        if constexpr (WPP == 1) asm volatile("" : "+s"(rec));  // (the record pointer of pairs_kernel is a scalar)
    }
    // the exact loop: the whole range of a pair without polynomials, the outer part of one whose range exceeds theirs
    float K0[8], K1[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        K0[i] = uniformf(rec->K0[i]);
        K1[i] = uniformf(rec->K1[i]);
    }
    const float dist_scale = n_t_f / p.range_t, dist_bias = fmaf(0.5f, n_t_f, 0.5f);
    if (reduce && p.quads) {
        // kappa_max > pi/4: in practice the pairs whose baseline passes through the object (kappa_max = pi/2).  Their
        // sampling curve crosses the whole Radon intermediate diagonally -- the 64 samples of a gather sit in ~17
        // different angle rows, one cache line each in the row-major copies -- and on their own they are memory-bound
        // (27.6 us per 1000 pairs against 8.5 us with the loads removed, scripts/exp_pair_classes.py).  In the row-quad
        // copy four consecutive rows share a line: 11.3 us per 1000 such pairs.  OPT-IN (ECC_QUAD_COPIES=1, 4x the slab
        // memory): inside the benchmark's mixed launch, where these are 3.5 % of the pairs, it buys 1 % (0.328 vs
        // 0.331 ms) -- there they cost 5 % as their own wave time and 5 % by slowing everybody else down
        // (scripts/exp_wave_timeline.py: a degree-8 wave takes 26.2 us next to them, 25.0 us without), whichever copy
        // they sample.  Useful for per-sample / index-list workloads made of such pairs.
        const SlabView q0 = {(GlobalBytes)p.quads[iD0], p.quad_group_bytes};
        const SlabView q1 = {(GlobalBytes)p.quads[iD1], p.quad_group_bytes};
        kappa_loop<DERIV, CORR, true, ECC_QUAD_LAYOUT, WPP>(k_first, p.k_limit, K0, K1, q0, q1, n_alpha_f, n_t_f, dist_scale, dist_bias,
                                                       pitch4_f, acc, mom2, mom3, mom4, stage);
    } else if (p.wide_offsets) {
        kappa_loop<DERIV, CORR, true, -1, WPP>(k_first, p.k_limit, K0, K1, sv0, sv1, n_alpha_f, n_t_f, dist_scale, dist_bias,
                                          pitch4_f, acc, mom2, mom3, mom4, stage);
    } else if (pitch4 == 6400u) {  // 768 distance bins, the reference's default (Gui/ComputeRadonIntermediate.hxx:43-44)
        if (reduce)
            kappa_loop<DERIV, CORR, true, 6400, WPP>(k_first, p.k_limit, K0, K1, sv0, sv1, n_alpha_f, n_t_f, dist_scale,
                                                dist_bias, pitch4_f, acc, mom2, mom3, mom4, stage);
        else
            kappa_loop<DERIV, CORR, false, 6400, WPP>(k_first, p.k_limit, K0, K1, sv0, sv1, n_alpha_f, n_t_f, dist_scale,
                                                 dist_bias, pitch4_f, acc, mom2, mom3, mom4, stage);
    } else {
        kappa_loop<DERIV, CORR, true, 0, WPP>(k_first, p.k_limit, K0, K1, sv0, sv1, n_alpha_f, n_t_f, dist_scale, dist_bias,
                                         pitch4_f, acc, mom2, mom3, mom4, stage);
    }
}


__device__ __forceinline__ float slab_texel(GlobalFloats slab, int pitch, int n_alpha, int n_t, int i, int j)
{
    i = min(max(i, 0), n_alpha - 1);
    j = min(max(j, 0), n_t - 1);
    return slab[(size_t)(i + 1) * pitch + (j + 1)];
}

// (a, d) in normalised texture coordinates -> value; W = n_alpha (x), H = n_t (y).
__device__ float slab_tex2d_norm(GlobalFloats slab, int pitch, int n_alpha, int n_t, float s, float t)
{
    const float x = s * (float)n_alpha, y = t * (float)n_t;
    const float xb = x - 0.5f, yb = y - 0.5f;
    const float fi = floorf(xb), fj = floorf(yb);
    const float fx = xb - fi, fy = yb - fj;
    // float -> int saturates on the device; the clamps below do the rest (d may be far outside [0, 1])
    const int i = (int)fmaxf(fminf(fi, 1e9f), -1e9f), j = (int)fmaxf(fminf(fj, 1e9f), -1e9f);
    const float T00 = slab_texel(slab, pitch, n_alpha, n_t, i, j), T10 = slab_texel(slab, pitch, n_alpha, n_t, i + 1, j);
    const float T01 = slab_texel(slab, pitch, n_alpha, n_t, i, j + 1),
                T11 = slab_texel(slab, pitch, n_alpha, n_t, i + 1, j + 1);
    const float r0 = (1.f - fx) * T00 + fx * T10;
    const float r1 = (1.f - fx) * T01 + fx * T11;
    return (1.f - fy) * r0 + fy * r1;
}

// ref: EpipolarConsistencyCommon.hxx:152-171 (lineToSampleDtr) + RadonIntermediate.h:86-105 (sample)
__device__ float sample_line_plain(const float* K, float x0, float x1, GlobalFloats slab, int pitch, int n_alpha,
                                   int n_t, float range_t, bool derivative, float* a_out, float* d_out)
{
    const float Pi = 3.14159265359f;
    float l0 = K[0] * x0 + K[3] * x1;
    float l1 = K[1] * x0 + K[4] * x1;
    float l2 = K[2] * x0 + K[5] * x1;
    const float length = sqrtf(l0 * l0 + l1 * l1);
    float a = (float)atan2((double)l1, (double)l0) / Pi;
    if (a < 0) a += 2;
    float d = -(l2 / length) / range_t + 0.5f;
    bool moved = false;
    if (a > 1) {
        a = a - 1.f;
        d = 1.f - d;
        moved = true;
    }
    *a_out = a;
    *d_out = d;
    const float v = slab_tex2d_norm(slab, pitch, n_alpha, n_t, a, d);
    return (derivative && moved) ? -v : v;
}

// ECC_SAMPLING_REFERENCE: the kappa samples first_k, first_k + stride, ... of one pair in the CPU path's own arithmetic
// (oracle/ecc_oracle.c or_pair / or_redundancy, i.e. ref: ...RadonIntermediate.cu:87-113,257-270 +
// EpipolarConsistencyCommon.hxx:152-171 as fp32 source expressions, sin / cos / atan2 through binary64 and rounded once,
// exact fp32 bilinear rule with index clamps on the dtr's own slab); float64 partial sums of this thread.
// STAGE: every sample's fp32 term(s) go to stage[k] (CORR: stage[k], stage[stage_stride + k], stage[2 * stage_stride + k])
// instead of into the sums -- the wide forms (1024 threads per pair) add them afterwards with reference_resum in the
// order of a 256-thread workgroup.
template <bool CORR, bool STAGE = false>
__device__ __forceinline__ void reference_loop(const EccPairParams& p, const float (&K0)[8], const float (&K1)[8],
                                               GlobalFloats d0, GlobalFloats d1, int first_k, int stride,
                                               double& acc, double& mom2, double& mom3, double& mom4,
                                               float* stage = nullptr, int stage_stride = 0)
{
    const float dkappa = K1[6], kappa_max = K1[7];
    const bool deriv = p.is_derivative != 0;
    for (int k = first_k; k < p.k_limit; k += stride) {
        const float kappa = dkappa * 0.5f + dkappa * k;  // ref: ...RadonIntermediate.cu:259
        if (kappa >= kappa_max) break;
        double sk, ck;
        sincos((double)kappa, &sk, &ck);  // the bits of sin() and cos(): one argument reduction, one pair of polynomials
        float x0 = (float)ck;
        const float x1 = (float)sk;
        float a, d;
        const float v0p = sample_line_plain(K0, x0, x1, d0, p.pitch, p.n_alpha, p.n_t, p.range_t, deriv, &a, &d);
        const float v1p = sample_line_plain(K1, x0, x1, d1, p.pitch, p.n_alpha, p.n_t, p.range_t, deriv, &a, &d);
        x0 *= -1;  // ref: ...RadonIntermediate.cu:106
        const float v0m = sample_line_plain(K0, x0, x1, d0, p.pitch, p.n_alpha, p.n_t, p.range_t, deriv, &a, &d);
        const float v1m = sample_line_plain(K1, x0, x1, d1, p.pitch, p.n_alpha, p.n_t, p.range_t, deriv, &a, &d);
        if (!CORR) {
            const float vp = v0p - v1p, vm = v0m - v1m;
            const float consistency = (vp * vp + vm * vm) * K0[6];  // ref: ...RadonIntermediate.cu:112
            const float term = consistency * dkappa;                // ref: ...RadonIntermediate.cu:269
            if (STAGE) stage[k] = term;
            else acc += (double)term;
        } else {
            const float one_over_n = kappa_max / kappa;  // ref: ...RadonIntermediate.cu:211,274
            const float t2 = one_over_n * (v0p * v0p + v0m * v0m), t3 = one_over_n * (v1p * v1p + v1m * v1m);
            const float t4 = one_over_n * (v0p * v1p + v0m * v1m);
            if (STAGE) {
                stage[k] = t2;
                stage[stage_stride + k] = t3;
                stage[2 * stage_stride + k] = t4;
            } else {
                mom2 += (double)t2;
                mom3 += (double)t3;
                mom4 += (double)t4;
            }
        }
    }
}

// The staged terms of reference_loop<CORR, true>, added the way thread T of a 256-thread workgroup accumulates them
// (k = T, T + 256, ... while kappa < kappa_max): the bits of pairs_reference_kernel<CORR, 4>'s per-thread sums.
template <bool CORR>
__device__ __forceinline__ void reference_resum(const EccPairParams& p, float dkappa, float kappa_max, int T, const float* stage,
                                                int stage_stride, double& acc, double& mom2, double& mom3, double& mom4)
{
    for (int k = T; k < p.k_limit; k += 256) {
        const float kappa = dkappa * 0.5f + dkappa * k;  // the fp32 operations of reference_loop
        if (kappa >= kappa_max) break;
        if (!CORR) {
            acc += (double)stage[k];
        } else {
            mom2 += (double)stage[k];
            mom3 += (double)stage[stage_stride + k];
            mom4 += (double)stage[2 * stage_stride + k];
        }
    }
}


}  // namespace

#endif
