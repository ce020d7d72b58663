// pairs_kernel.hip -- all-pairs / index-list epipolar-consistency kernel for gfx950.
//
// Two stream-ordered launches do what the reference needs two kernels, two device-wide syncs and N_kappa
// float atomics per pair for (ref: LibEpipolarConsistency/EpipolarConsistencyRadonIntermediate.cu):
//   kernelEpipolarConsistencyComputeK01 (:13-67)  -> k01_kernel: pair geometry plus, per view of the pair, the
//                                                    sample coordinates as polynomials in kappa (EccPairRecord,
//                                                    296 bytes), read by the pair kernel with scalar loads;
//   kernelEpipolarCosistency<deriv,false> (:152-276) -> pairs_kernel: one wave64 per pair, kappa samples strided
//                                                    over the lanes, shuffle reduction, ONE plain store.
// Sampling replaces tex2D on a normalised, clamped, bilinear texture (ref: RadonIntermediate.cpp:192)
// by the exact fp32 rule of SURVEY.md 8c on a row-paired copy of the padded, distance-fast slab of
// ecc_layout.h: ONE 16-byte load per sample (the 2x2 footprint is contiguous), no index clamps.
// The per-pair sum is carried in binary64 (the reference's atomicAdd order is arbitrary; the oracle
// does the same), so results do not depend on the reduction tree.
#include <hip/hip_runtime.h>
#include <float.h>

#include "ecc_layout.h"
#include "ecc_pairs_device.h"

namespace {


// Pair geometry and polynomial fit (ref for the geometry: kernelEpipolarConsistencyComputeK01,
// ...RadonIntermediate.cu:13-67).  A wave of the pair kernel would spend ~900 vector instructions (float64 asin, ~25
// IEEE divisions, get_ij) on the geometry per pair with all 64 lanes doing the same thing; here 64 pairs share those
// instructions and the pair kernel picks the result up with scalar loads.  Unlike the reference there is no
// device-wide sync in between, the two kernels are just ordered on the stream.
// FOUR threads per pair, one per (view, coordinate), arranged so that a wave has ONE role (no divergence between the
// angle and the distance code): workgroup = 64 pairs; wave 0 / 1 = angle coordinate of view 0 / 1, wave 2 / 3 =
// distance coordinate of view 0 / 1.  Every thread computes its view's half of K01 itself (cheaper than passing it
// on).  A thread is a chain of dependent float64 operations, and a shard of the pair range at 8 GPUs is less than one
// wave per SIMD -- the chain length is the kernel time there.  Measured: 79 800 pairs 35 us with float64 libm-style
// series and divisions, 22 us with sincos_table / angle_table and explicit fma (both two threads per pair), 23 us
// with four; a 9 975-pair shard (8 GPUs) 13 us with two threads per pair, 11 us with four.

// LANES: threads per (pair, role) -- 1: one thread does the whole fit (throughput form, 64 pairs per workgroup); 8: the
// nodes and checks of a fit are spread over 8 adjacent lanes (fit_coordinate_wide, 8 pairs per workgroup) and everything
// else is done redundantly by all of them: 2.6x the instructions in total, a third of the chain length -- the form for
// launches of a few thousand pairs (index lists, pose-delta launches, the shard of one of 8 GPUs), where the kernel's
// time IS the chain length.  Identical records either way.
// The records of the workgroup's pairs (k01_fit_block) out of LDS into p.records.
template <int LANES>
__device__ __forceinline__ void k01_block(const EccPairParams& p, K01Shared<LANES>& sh, EccSmallEvalArg small, bool poison)
{
    constexpr int K01_PAIRS = 64 / LANES;
    // The records are assembled in LDS and leave the workgroup as one contiguous, coalesced block: written straight
    // from the threads every store instruction would scatter its 64 lanes over 64 records 296 bytes apart.
    k01_fit_block<LANES>(p, (long long)blockIdx.x * K01_PAIRS, K01_PAIRS, sh, small);
    const long long first_pair = (long long)blockIdx.x * K01_PAIRS;
    const long long n_here = min((long long)K01_PAIRS, p.count - first_pair);
    if (n_here > 0) {
        constexpr int RW = (int)(sizeof(EccPairRecord) / 8);
        const int words = (int)n_here * RW;
        if (poison) {  // the kernel's two views of its argument list disagree: every value of the launch becomes NaN
            if (threadIdx.x < n_here) sh.recs[threadIdx.x].K0[6] = __builtin_nanf("");
            __syncthreads();
        }
        const double* src = reinterpret_cast<const double*>(sh.recs);
        if (!p.record_slots) {
            double* dst = reinterpret_cast<double*>(p.records + first_pair);
            for (int q = threadIdx.x; q < words; q += 256) dst[q] = src[q];
        } else {  // each record into its own slot (still whole records, 37 consecutive 8-byte words)
            for (int q = threadIdx.x; q < words; q += 256) {
                const int rcd = q / RW, w = q - rcd * RW;
                reinterpret_cast<double*>(p.records + p.record_slots[p.first + first_pair + rcd])[w] = src[q];
            }
        }
    }
}

// The wide forms (LANES > 1: the few hundred pairs of a moved view, refitted on the side stream while the all-pairs launch runs)
// must fit the hole ONE retiring workgroup of pairs_kernel leaves on its compute unit: 72 + 8 vector registers per SIMD.  At 79
// registers k01_kernel<16> is through in 45 us beside the big launch and the moved pairs' own launch follows beside it as well; at
// 126 -- what the compiler chose after a change elsewhere in this file -- it found no room until the big launch had drained
// (283 us, kernel trace of bench.py) and the step paid for the refit and the list launch in full (+15 us).  (Forcing 80 with
// amdgpu_num_vgpr(40) makes that version spill 212 bytes; writing the change so that the compiler stays at 79 by itself does
// not.)  tests/test_kernel_resources.py fails the CPU suite when a build leaves the budget.
template <int LANES>
__global__ __launch_bounds__(256) void k01_kernel(EccPairParams p)
{
    __shared__ K01Shared<LANES> sh;
    k01_block<LANES>(p, sh, nullptr, false);
}

// k01_kernel<8> with E1 of the views whose matrix changed since the device arrays were made in the KERNEL ARGUMENTS (x.patch_*,
// at most ECC_SMALL_PATCH_MAX views, computed by the host with e1_kernel's own code): an optimiser step that moves a few
// views needs no e1_kernel launch in front of a small evaluation.  The list is read in place in the kernel-argument segment,
// as in small_eval_kernel.hip; workgroup 0 stores the entries into the device arrays for later launches.
template <int LANES>
__global__ __launch_bounds__(256) void k01_patched_kernel(EccPairParams p, EccSmallEval x)
{
    __shared__ K01Shared<LANES> sh;
    typedef const char __attribute__((address_space(4))) * KernargBytes;
    static_assert(alignof(EccSmallEval) == 8 && alignof(EccPairParams) == 8, "layout of the kernel arguments");
    const EccSmallEvalArg xs = (EccSmallEvalArg)((KernargBytes)__builtin_amdgcn_kernarg_segment_ptr() + ((sizeof(EccPairParams) + 7) & ~(size_t)7));
    const bool args_ok = xs->magic == ECC_SMALL_MAGIC && x.magic == ECC_SMALL_MAGIC && xs->patch_count == x.patch_count;
    k01_block<LANES>(p, sh, xs, !args_ok);
}

// Seven waves per SIMD instead of six (round 5).  The hardware keeps 16 scalar registers per wave for the trap handler on top
// of the kernel's own allocation: measured on MI355X with 256-thread workgroups (scripts/micro/sgpr_occupancy.hip, wave-slot ids
// of HW_REG_HW_ID) an allocation of up to 80 registers runs 8 waves per SIMD, up to 96 7, above that 6 -- one less than the
// compiler's and the runtime's own estimate.  Left alone this kernel allocates 112 (100 + VCC + 4, rounded to 8) and the time of
// a launch follows its occupancy steeply (workgroups per CU capped with dynamic LDS: 3 / 4 / 5 / 6 -> 0.74 / 0.426 / 0.370 /
// 0.329 ms).  With the allocation capped at 96 the compiler moves 40 scalar values into vector-register lanes (v_writelane /
// v_readlane), every one of them outside the sampling loops: 0.329 -> 0.320 ms, 2 850-2 870 -> 2 935 evaluations/s A/B/A/B on one
// box, the same bits.  A cap of 80 (8 waves) puts 258 such moves into the loops as well and ends at the same 0.320 ms.
// (Since the exact loop reads its K0 / K1 only after the polynomial loops -- pair_accumulate -- the kernel needs 88 and nothing
// is moved into lanes any more.)
// Seven waves also need at most 72 vector registers (7 x 72 <= 512); the two-steps-per-trip loop sits right at that line (72 or 77
// from one edit to the next), so it is pinned: amdgpu_num_vgpr counts BOTH halves of gfx950's unified register file and a kernel
// without accumulation registers gets all of it as vector registers -- 36 means 72 (measured: 72 is accepted and changes nothing,
// 36 gives 72 without scratch).  amdgpu_waves_per_eu(7, 8) reaches 65 registers and a 0.51 ms launch (0.305 without), on the
// kernel of the round before this change as well: do not use it.  tests/test_kernel_resources.py reads the numbers back from
// the built library.
#ifndef PK_OCCUPANCY
#define PK_OCCUPANCY __attribute__((amdgpu_num_sgpr(96), amdgpu_num_vgpr(36)))
#endif
template <bool DERIV, bool CORR>
__global__ __launch_bounds__(PK_MAIN_THREADS) PK_OCCUPANCY void pairs_kernel(EccPairParams p)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Workgroup -> pairs.  XCD-aware: workgroups b and b+8 share an XCD, and an XCD walks a contiguous part of the
    // pair order (consecutive pairs share view i and have neighbouring partners j: their dtr bands are re-used out
    // of that XCD's L2).  The four waves of a workgroup, however, take pairs a quarter of the range apart: with four
    // CONSECUTIVE pairs per workgroup the waves run in lockstep through nearly the same lines of view i, and the L1
    // serialises hits on lines whose fill is still in flight (TCP_PENDING_STALL_CYCLES: a quarter of its busy time)
    // -- 0.400 -> 0.335 ms for 79 800 pairs; spreading further (other strides, 2-D tiles of views that halve the HBM
    // traffic) was slower, see DESIGN.md 4.2.
    // (Several waves per pair for small shards -- wave h takes the kappa iterations it % split == h, float64 partial
    // sums combined by the sum kernel -- were measured and dropped: a 9 975-pair shard 80 us per step with whole-pair
    // waves, 89 us with two, 106 us with four waves per pair.)
    // (A host-made launch schedule that gives the kappa_max = pi/2 pairs evenly spaced positions, everything else keeping
    // its order: with explicit index lists over all 79 800 pairs it looked promising -- natural order 0.361 ms, those
    // pairs first 0.460, last 0.422, as whole workgroups first 0.601, spread evenly 0.348, scripts/exp_heavy_first.py --
    // but built into this mapping it was 1.5 % SLOWER, 0.3365 vs 0.3315 ms: the four-quarters mapping already spreads
    // them over the first quarter of every workgroup.)
    // (Persistent waves -- a launch sized to be resident at once, every wave handling several pairs in turn -- were
    // measured too: 79 800 pairs 0.38 / 0.43 ms with 5 / 10 pairs per wave against 0.33 ms, the shard 86 us with two.)
    const long long nblk = (p.count + PK_MAIN_WAVES - 1) / PK_MAIN_WAVES;
    const long long per_xcd = (nblk + 7) / 8;
    const long long blk = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (blk >= nblk) return;
    long long local = (long long)wave * nblk + blk;
    if (local >= p.count) return;  // no barriers below: waves leave independently
    // wave-uniform record -> SGPRs (readfirstlane makes the address provably uniform: scalar loads)
    local = ((long long)__builtin_amdgcn_readfirstlane((int)(local >> 32)) << 32) |
            (unsigned)__builtin_amdgcn_readfirstlane((int)local);
    // a list launch over records kept in their slots of an all-pairs array (record reuse): the slot comes from the list
    const long long rec_index = p.record_slots ? (long long)__builtin_amdgcn_readfirstlane(p.record_slots[local]) : local;
    const EccPairRecord* __restrict__ rec = p.records + rec_index;
    const int iD0 = __builtin_amdgcn_readfirstlane(rec->iD0), iD1 = __builtin_amdgcn_readfirstlane(rec->iD1);
    if (p.skip_enabled) {  // this pair is being refitted and sampled by the list launch on the second stream
        const unsigned m0 = p.skip_mask[(iD0 >> 5) & (ECC_SKIP_WORDS - 1)], m1 = p.skip_mask[(iD1 >> 5) & (ECC_SKIP_WORDS - 1)];
        if (((m0 >> (iD0 & 31)) | (m1 >> (iD1 & 31))) & 1u) return;
    }
    const int ci = __builtin_amdgcn_readfirstlane(rec->ci), cj = __builtin_amdgcn_readfirstlane(rec->cj);
    double acc = 0.0;
    double mom2 = 0.0, mom3 = 0.0, mom4 = 0.0;  // CORR: sums of x*x, y*y, x*y (x, y alone are not used by cc())
    pair_accumulate<DERIV, CORR>(p, rec, iD0, iD1, lane, acc, mom2, mom3, mom4);
    float val;
    if (!CORR) {
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
        val = (float)acc;
    } else {
        for (int off = 32; off > 0; off >>= 1) {
            mom2 += __shfl_down(mom2, off);
            mom3 += __shfl_down(mom3, off);
            mom4 += __shfl_down(mom4, off);
        }
        // host epilogue of the reference: cc = xy / (sqrt(xx) sqrt(yy)), cost = (1 - cc) * weight(= 1)
        // (ref: ...RadonIntermediate.cpp:127-131,199-210)
        const float xx = (float)mom2, yy = (float)mom3, xy = (float)mom4;
        const float corr = (float)((double)xy / (sqrt((double)xx) * sqrt((double)yy)));
        val = (1.0f - corr) * 1.0f;
    }
    if (lane == 0) {
        if (p.pair_values) p.pair_values[p.value_slots ? (long long)p.value_slots[local] : local] = val;
        if (p.cost && !p.indices) p.cost[(size_t)ci + (size_t)cj * p.n_views] = val;
    }
    // (Fusing the final float64 sum in here -- last-ticket wave reduces -- was measured and dropped: one
    // device-scope atomic per wave on a single counter serialises, 0.51 -> 1.05 ms, and with acquire/release
    // fences per wave 3.15 ms: a cross-XCD release writes the L2 back.  The sum stays a 8-us kernel of its own.)
}

// Launches of at most ECC_PAIRS_SPLIT_MAX pairs (small all-pairs evaluations, shards of them, index lists, the moved view's
// pairs of the pose-delta mode): one wave per pair leaves most SIMDs idle and the launch lasts as long as one wave's 12 - 23
// dependent trips.  Here WPP = 4 or 2 waves share a pair: wave `sub` takes the 64-sample trips sub, sub + WPP, ... and
// stores every sample's term in LDS; the pair's first wave then adds the terms per lane in the order ONE wave accumulates
// them (k = lane, lane + 64, ... while kappa < kappa_max), so the value has pairs_kernel's bits
// (tests/test_gpu_small_eval.py: the same pairs through both kernels).  Round 4, config 2 (2016 pairs): 16.9 -> ~7 us.
// WPP = 8 (one pair per 512-thread workgroup, launches of at most ECC_PAIRS_SPLIT8_MAX pairs): a 512-pair index list
// 30.2 -> 28.7 us end to end, smaller lists and the pose-delta sweep unchanged (their time is the record kernel and the
// launches around it).
template <bool DERIV, int WPP>
__global__ __launch_bounds__(WPP > 4 ? 64 * WPP : PK_THREADS) void pairs_split_kernel(EccPairParams p, int stage_stride)
{
    constexpr int PPW = WPP > 4 ? 1 : 4 / WPP;  // pairs per workgroup (WPP = 8: one pair per 512-thread workgroup)
    extern __shared__ float stage_all[];  // PPW * stage_stride floats
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = wave / WPP, sub = wave % WPP;  // wave-uniform
    long long local = (long long)blockIdx.x * PPW + slot;
    const bool live = local < p.count;
    local = ((long long)__builtin_amdgcn_readfirstlane((int)(local >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)local);
    double acc = 0.0, m2 = 0.0, m3 = 0.0, m4 = 0.0;
    const EccPairRecord* __restrict__ rec = nullptr;
    float* stage = stage_all + (size_t)slot * stage_stride;
    if (live) {
        const long long rec_index = p.record_slots ? (long long)__builtin_amdgcn_readfirstlane(p.record_slots[local]) : local;
        rec = p.records + rec_index;
        const int iD0 = __builtin_amdgcn_readfirstlane(rec->iD0), iD1 = __builtin_amdgcn_readfirstlane(rec->iD1);
        pair_accumulate<DERIV, false, WPP>(p, rec, iD0, iD1, lane, acc, m2, m3, m4, sub, stage);
    }
    __syncthreads();  // every wave of the pair has stored its trips
    if (!live || sub != 0) return;
    const float dkappa = uniformf(rec->K1[6]), kappa_max = uniformf(rec->K1[7]);
    for (int k = lane; k < p.k_limit; k += 64) {
        const float kappa = dkappa * 0.5f + dkappa * k;  // ref: ...RadonIntermediate.cu:259 (same fp32 ops)
        if (kappa >= kappa_max) break;
        acc += (double)stage[k];
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0) {
        const float val = (float)acc;
        if (p.pair_values) p.pair_values[p.value_slots ? (long long)p.value_slots[local] : local] = val;
        if (p.cost && !p.indices) {
            const int ci = rec->ci, cj = rec->cj;
            p.cost[(size_t)ci + (size_t)cj * p.n_views] = val;
        }
    }
}

// Deterministic float64 sum of `count` pair values (single workgroup, fixed tree).
// ref: ...RadonIntermediate.cpp:216-224 (host loop; weights are all 1).
// values_host (optional, pinned and device-mapped): the kernel also hands the values themselves to the host -- every thread
// stores what it loads, 16 contiguous bytes per lane, drained before the barrier in front of the result's store; writes of
// one device to host memory arrive in order, so a host that sees the result sees the values (index lists: no copy command).
__global__ __launch_bounds__(1024) void sum_pairs_kernel(const float* __restrict__ vals, long long count,
                                                         double* __restrict__ out, float* __restrict__ values_host)
{
    __shared__ double s[1024 / 64];
    // 16-byte loads, four independent float64 accumulators per thread (fixed order => deterministic)
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const long long n4 = count >> 2;
    const float4* __restrict__ v4 = reinterpret_cast<const float4*>(vals);
    if (values_host) {  // uniform; its own pass, so that the arithmetic below is the one code path it always was
        // system-scope stores (plain stores to host memory may sit in the L2 until the kernel ends; the host reads the
        // values as soon as it sees the result)
        for (long long q = threadIdx.x; q < count; q += 1024)
            __hip_atomic_store(reinterpret_cast<unsigned*>(values_host) + q, __float_as_uint(vals[q]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // eight loads in flight per thread: the kernel is one workgroup reading what other XCDs have just written
    // (HBM latency each time), issued one at a time it took 10 us for 79 800 values
    long long k = threadIdx.x;
    // (All of a thread's ~20 loads in flight at once would make it one round trip, but 1024 threads x 20 float4 do not fit
    // the 128 registers a thread of this workgroup may have: it spilled.  Ten per batch = two round trips.)
    for (; k + 9 * 1024 < n4; k += 10 * 1024) {
        float4 v[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) v[u] = v4[k + u * 1024];
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            a0 += (double)v[u].x;
            a1 += (double)v[u].y;
            a2 += (double)v[u].z;
            a3 += (double)v[u].w;
        }
    }
    for (; k + 7 * 1024 < n4; k += 8 * 1024) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = v4[k + u * 1024];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 += (double)v[u].x;
            a1 += (double)v[u].y;
            a2 += (double)v[u].z;
            a3 += (double)v[u].w;
        }
    }
    for (; k < n4; k += 1024) {
        const float4 v = v4[k];
        a0 += (double)v.x;
        a1 += (double)v.y;
        a2 += (double)v.z;
        a3 += (double)v.w;
    }
    double acc = (a0 + a1) + (a2 + a3);
    if (threadIdx.x == 0)
        for (long long k = n4 << 2; k < count; ++k) acc += (double)vals[k];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < 1024 / 64; w++) tot += s[w];
        // `out` is usually pinned host memory that the host polls (ecc_capi.hip: wait_result): one 8-byte store at
        // system scope, written through, visible to the host before the kernel's end-of-dispatch write-back
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(tot),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// -------------------------------------------------------------------------------------------------
// E7: evaluateForImagePair -- the two redundant signals of ONE pair over kappa in (-kappa_max, kappa_max),
// for plotting (ref: EpipolarConsistencyRadonIntermediate.cpp:324-393, "visualization only").  One thread
// per kappa sample; this is the oracle's plain arithmetic (or_pair_samples in oracle/ecc_oracle.c): the
// reference's float expressions in source order, elementary functions correctly rounded, bilinear rule of
// SURVEY.md 8c with explicit index clamps -- not the fast path of pairs_kernel.
// Deviations from the reference's host code, which is visibly unfinished: the fold's sign is
// applied (its sample() can never reach the flip after lineToSampleDtr folded the angle,
// RadonIntermediate.h:91-100), and sampling uses the texel rule of the metric itself (a*n_alpha - .5) instead
// of the host image's (n-1)*s scaling (RadonIntermediate.h:108), so the curves are what evaluate() compares.
// -------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void pair_samples_kernel(EccPairSamplesParams p)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    float K0[8], K1[8];
    if (p.iP0 == p.iP1) {
        for (int i = 0; i < 8; i++) K0[i] = K1[i] = 0.f;
    } else {
        compute_K01(p.n_x2, p.n_y2, p.Cs + 4 * p.iP0, p.Cs + 4 * p.iP1, p.PinvTs + 12 * p.iP0, p.PinvTs + 12 * p.iP1,
                    p.object_radius_mm, p.num_samples, p.dkappa_user, true, K0, K1);
    }
    if (k == 0)
        for (int i = 0; i < 8; i++) {
            p.K01_out[i] = K0[i];
            p.K01_out[8 + i] = K1[i];
        }
    const float dkappa = K1[6], kappa_max = K1[7];
    // ref: ...RadonIntermediate.cpp:367: for (float kappa=-kappa_max+0.5f*dkappa; kappa<kappa_max; kappa+=dkappa)
    // -- kappa accumulates in fp32; thread k replays the first k additions.
    float kappa = -kappa_max + 0.5f * dkappa;
    if (!(dkappa > 0.f)) return;  // degenerate pair: the reference's loop would not terminate / not start
    for (int q = 0; q < k && kappa < kappa_max; ++q) kappa += dkappa;
    if (!(kappa < kappa_max) || k >= p.capacity) return;
    if (!(kappa + dkappa < kappa_max) || k + 1 == p.capacity) *p.n_out = k + 1;  // exactly one thread: the last sample
    const float x0 = (float)cos((double)kappa), x1 = (float)sin((double)kappa);
    float a0, d0, a1, d1;
    const float v0 = sample_line_plain(K0, x0, x1, (GlobalFloats)p.dtr0, p.pitch, p.n_alpha, p.n_t, p.range_t, p.derivative0 != 0, &a0, &d0);
    const float v1 = sample_line_plain(K1, x0, x1, (GlobalFloats)p.dtr1, p.pitch, p.n_alpha, p.n_t, p.range_t, p.derivative1 != 0, &a1, &d1);
    p.out[0 * (size_t)p.capacity + k] = v0;
    p.out[1 * (size_t)p.capacity + k] = v1;
    p.out[2 * (size_t)p.capacity + k] = kappa;
    p.out[3 * (size_t)p.capacity + k] = a0;
    p.out[4 * (size_t)p.capacity + k] = d0;
    p.out[5 * (size_t)p.capacity + k] = a1;
    p.out[6 * (size_t)p.capacity + k] = d1;
}

// -------------------------------------------------------------------------------------------------
// ECC_SAMPLING_REFERENCE: the pair loop in the CPU path's own arithmetic (oracle/ecc_oracle.c or_pair / or_redundancy,
// i.e. ref: ...RadonIntermediate.cu:87-113,257-270 + EpipolarConsistencyCommon.hxx:152-171 as fp32 source
// expressions, sin/cos/atan2 through binary64 and rounded once, exact fp32 bilinear rule with index clamps on the
// dtr's own slab).  One wave per pair like pairs_kernel, kappa samples strided over the lanes, per-lane float64
// partial sums (the only difference to a sequential CPU loop: the order of a float64 sum).  ~10x the instructions of
// the polynomial path per sample -- meant for evaluations of few pairs, where a single pair's value has to agree with
// the CPU path (include/ecc_hip.h, ecc_metric_set_sampling).
// -------------------------------------------------------------------------------------------------
// SPLIT = 4: the four waves of a workgroup share ONE pair (thread T takes k = T, T + 256, ...; the wave sums are added in
// wave order through LDS) -- a launch of at most a few hundred pairs leaves most of the 1024 SIMDs idle, and the time of
// the evaluation is the time of one pair: 81 -> ~50 us for a single-pair evaluation.  The float64 partial sums are
// grouped differently from SPLIT = 1, so the mode of a metric is fixed by the size of the FULL range it evaluates
// (fill_pair_params), not by the launch: pose-delta launches reproduce the full evaluation's bits.
template <bool CORR, int SPLIT>
__global__ __launch_bounds__(PK_THREADS) void pairs_reference_kernel(EccPairParams p)
{
    static_assert(SPLIT == 1 || SPLIT == PK_THREADS / 64, "one pair per wave or per workgroup");
    __shared__ double part[3][PK_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long local = SPLIT == 1 ? (long long)blockIdx.x * 4 + wave : (long long)blockIdx.x;
    if (local >= p.count) return;  // SPLIT > 1: uniform over the workgroup
    local = ((long long)__builtin_amdgcn_readfirstlane((int)(local >> 32)) << 32) |
            (unsigned)__builtin_amdgcn_readfirstlane((int)local);
    const EccPairRecord* __restrict__ rec = p.records + local;
    float K0[8], K1[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        K0[i] = uniformf(rec->K0[i]);
        K1[i] = uniformf(rec->K1[i]);
    }
    const int iD0 = __builtin_amdgcn_readfirstlane(rec->iD0), iD1 = __builtin_amdgcn_readfirstlane(rec->iD1);
    const int ci = __builtin_amdgcn_readfirstlane(rec->ci), cj = __builtin_amdgcn_readfirstlane(rec->cj);
    const GlobalFloats d0 = (GlobalFloats)p.slabs[iD0];
    const GlobalFloats d1 = (GlobalFloats)p.slabs[iD1];
    double acc = 0.0, mom2 = 0.0, mom3 = 0.0, mom4 = 0.0;
    reference_loop<CORR>(p, K0, K1, d0, d1, SPLIT == 1 ? lane : (int)threadIdx.x, 64 * SPLIT, acc, mom2, mom3, mom4);
    if (!CORR) {
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    } else {
        for (int off = 32; off > 0; off >>= 1) {
            mom2 += __shfl_down(mom2, off);
            mom3 += __shfl_down(mom3, off);
            mom4 += __shfl_down(mom4, off);
        }
    }
    if (SPLIT > 1) {  // wave sums -> wave 0, added in wave order
        if (lane == 0) {
            part[0][wave] = CORR ? mom2 : acc;
            part[1][wave] = mom3;
            part[2][wave] = mom4;
        }
        __syncthreads();
        if (wave != 0) return;
        acc = mom2 = part[0][0];
        mom3 = part[1][0];
        mom4 = part[2][0];
#pragma unroll
        for (int w = 1; w < PK_THREADS / 64; ++w) {
            acc += part[0][w];
            mom2 += part[0][w];
            mom3 += part[1][w];
            mom4 += part[2][w];
        }
    }
    float val;
    if (!CORR) {
        val = (float)acc;
    } else {
        const float xx = (float)mom2, yy = (float)mom3, xy = (float)mom4;
        const float corr = (float)((double)xy / (sqrt((double)xx) * sqrt((double)yy)));
        val = (1.0f - corr) * 1.0f;
    }
    if (lane == 0) {
        if (p.pair_values) p.pair_values[p.value_slots ? (long long)p.value_slots[local] : local] = val;
        if (p.cost && !p.indices) p.cost[(size_t)ci + (size_t)cj * p.n_views] = val;
    }
}

// The wide form of pairs_reference_kernel<CORR, 4> for launches that leave most of the chip idle (at most
// ECC_REFERENCE_WIDE_MAX_PAIRS pairs: two such workgroups per CU hold them all at once): 1024 threads share ONE pair, thread
// T takes k = T, T + 1024, ... and stores each sample's fp32 term(s) in LDS; the first 256 threads then add them in the
// order of the 256-thread kernel (thread T: k = T, T + 256, ...; wave sums in wave order), so every pair value has its
// bits.  The binary64 sin / cos / atan2 chains of a sample are a few hundred dependent instructions: at one wave per SIMD
// they run at a quarter of the issue rate, and a pair's 1448 samples took six trips per thread instead of two.
#define ECC_REFERENCE_WIDE_THREADS 1024
template <bool CORR>
__global__ __launch_bounds__(ECC_REFERENCE_WIDE_THREADS) void pairs_reference_wide_kernel(EccPairParams p, int stage_stride)
{
    extern __shared__ float ref_stage[];  // (CORR ? 3 : 1) * stage_stride floats
    __shared__ double part[3][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long local = (long long)blockIdx.x;  // < p.count by the launch
    const EccPairRecord* __restrict__ rec = p.records + local;
    float K0[8], K1[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        K0[i] = uniformf(rec->K0[i]);
        K1[i] = uniformf(rec->K1[i]);
    }
    const int iD0 = __builtin_amdgcn_readfirstlane(rec->iD0), iD1 = __builtin_amdgcn_readfirstlane(rec->iD1);
    const int ci = __builtin_amdgcn_readfirstlane(rec->ci), cj = __builtin_amdgcn_readfirstlane(rec->cj);
    double acc = 0.0, mom2 = 0.0, mom3 = 0.0, mom4 = 0.0;
    reference_loop<CORR, true>(p, K0, K1, (GlobalFloats)p.slabs[iD0], (GlobalFloats)p.slabs[iD1], (int)threadIdx.x,
                               ECC_REFERENCE_WIDE_THREADS, acc, mom2, mom3, mom4, ref_stage, stage_stride);
    __syncthreads();
    if (wave < 4) {  // wave-uniform
        reference_resum<CORR>(p, K1[6], K1[7], (int)threadIdx.x, ref_stage, stage_stride, acc, mom2, mom3, mom4);
        if (!CORR) {
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
        } else {
            for (int off = 32; off > 0; off >>= 1) {
                mom2 += __shfl_down(mom2, off);
                mom3 += __shfl_down(mom3, off);
                mom4 += __shfl_down(mom4, off);
            }
        }
        if (lane == 0) {
            part[0][wave] = CORR ? mom2 : acc;
            part[1][wave] = mom3;
            part[2][wave] = mom4;
        }
    }
    __syncthreads();
    if (wave != 0) return;
    acc = mom2 = part[0][0];
    mom3 = part[1][0];
    mom4 = part[2][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        acc += part[0][w];
        mom2 += part[0][w];
        mom3 += part[1][w];
        mom4 += part[2][w];
    }
    float val;
    if (!CORR) {
        val = (float)acc;
    } else {
        const float xx = (float)mom2, yy = (float)mom3, xy = (float)mom4;
        const float corr = (float)((double)xy / (sqrt((double)xx) * sqrt((double)yy)));
        val = (1.0f - corr) * 1.0f;
    }
    if (lane == 0) {
        if (p.pair_values) p.pair_values[p.value_slots ? (long long)p.value_slots[local] : local] = val;
        if (p.cost && !p.indices) p.cost[(size_t)ci + (size_t)cj * p.n_views] = val;
    }
}

}  // namespace

extern "C" hipError_t ecc_launch_pair_samples(const EccPairSamplesParams* p, hipStream_t stream)
{
    hipLaunchKernelGGL(pair_samples_kernel, dim3((p->capacity + 255) / 256), dim3(256), 0, stream, *p);
    return hipGetLastError();
}

namespace {

// Row-paired copy of a dtr for the pair kernel: element (row r, bin j, h) = slab(r + h, j), i.e. the four taps of a
// bilinear footprint are 16 contiguous bytes and one global_load_dwordx4 fetches them.  The vector memory path
// takes ~16 cycles per wave-level gather instruction whatever its width, and with the coordinates coming from
// polynomials the two 8-byte loads per sample had become the kernel's bottleneck (measured: 0.50 ms with them,
// 0.36 ms with one, 0.33 ms with none).  Costs a second, twice as large copy of the stack in HBM (2 GB of 288).
__global__ __launch_bounds__(256) void build_paired_kernel(const float* const* __restrict__ slabs, float* __restrict__ paired,
                                                           int64_t paired_stride, int rows, int pitch)
{
    const float* __restrict__ s = slabs[blockIdx.z];
    float* __restrict__ d = paired + (int64_t)blockIdx.z * paired_stride;
    const int r = blockIdx.y;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < pitch; j += gridDim.x * blockDim.x) {
        // (value of row r, row r+1 minus row r): the fp32 difference the bilinear rule needs is formed here once instead
        // of in every sample -- the same rounded number, so results do not change (scripts/pair_values_digest.py: the
        // SHA-256 of all 79 800 pair values is the same in both modes); two vector instructions less per sample,
        // 0.339 -> 0.333 ms A/B on one box
        const float a = s[(size_t)r * pitch + j];
        F2 v = {a, s[(size_t)(r + 1) * pitch + j] - a};
        *reinterpret_cast<F2*>(d + ((size_t)r * pitch + j) * 2) = v;
    }
}

}  // namespace

namespace {
// Row-quad copy of a dtr for the pairs that sweep many angle rows per gather (see pairs_kernel): per group g of four
// padded rows and bin j four float4 footprints (row 4g+q, bin j) = { S(r,j), S(r+1,j), S(r,j+1), S(r+1,j+1) }, q = 0..3,
// so that one 128-byte line holds 4 rows x 2 bins.  rows = n_alpha + 1 footprint rows, groups = ceil(rows / 4); rows
// past the slab repeat its last row pair (never sampled).  4x the slab's size (3.9 GB for 400 views of 768 x 768 bins).
__global__ __launch_bounds__(256) void build_quad_kernel(const float* const* __restrict__ slabs, F4* __restrict__ quads,
                                                         int64_t quad_stride_f4, int rows, int pitch)
{
    const float* __restrict__ s = slabs[blockIdx.z];
    F4* __restrict__ d = quads + (int64_t)blockIdx.z * quad_stride_f4;
    const int g = blockIdx.y;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < pitch * 4; e += gridDim.x * blockDim.x) {
        const int j = e >> 2, q = e & 3;
        const int r = min(4 * g + q, rows - 1);
        const int j1 = min(j + 1, pitch - 1);
        const float a = s[(size_t)r * pitch + j], b = s[(size_t)r * pitch + j1];
        const F4 v = {a, s[(size_t)(r + 1) * pitch + j] - a, b, s[(size_t)(r + 1) * pitch + j1] - b};  // value, row difference
        d[(size_t)g * pitch * 4 + e] = v;
    }
}
}  // namespace

extern "C" hipError_t ecc_launch_build_quad(const float* const* slabs_tbl_d, float* quads_d, int64_t quad_stride_floats, int n,
                                            int rows, int pitch, hipStream_t stream)
{
    dim3 grid((pitch * 4 + 255) / 256, (rows + 3) / 4, n);
    hipLaunchKernelGGL(build_quad_kernel, grid, dim3(256), 0, stream, slabs_tbl_d, reinterpret_cast<F4*>(quads_d),
                       quad_stride_floats / 4, rows, pitch);
    return hipGetLastError();
}

// slabs_tbl_d: device table of n slab pointers (private layout, rows+1 rows each); paired_d: n paired copies of
// `rows` x pitch x 2 floats each, rows = n_alpha + 1.
extern "C" hipError_t ecc_launch_build_paired(const float* const* slabs_tbl_d, float* paired_d, int64_t paired_stride, int n,
                                              int rows, int pitch, hipStream_t stream)
{
    dim3 grid((pitch + 255) / 256, rows, n);
    hipLaunchKernelGGL(build_paired_kernel, grid, dim3(256), 0, stream, slabs_tbl_d, paired_d, paired_stride, rows, pitch);
    return hipGetLastError();
}

extern "C" hipError_t ecc_launch_k01(const EccPairParams* p, hipStream_t stream)
{
    if (p->count <= 0) return hipSuccess;
    // small launches: 8 lanes per fit (the kernel's time is one thread's chain there); the records are identical
    if (p->count <= ECC_K01_LANES16_MAX_PAIRS) {
        constexpr int per_wg = 64 / ECC_K01_SMALL_LANES;
        hipLaunchKernelGGL(k01_kernel<ECC_K01_SMALL_LANES>, dim3((unsigned)((p->count + per_wg - 1) / per_wg)), dim3(256), 0, stream, *p);
    } else if (p->count <= ECC_K01_WIDE_MAX_PAIRS)
        hipLaunchKernelGGL(k01_kernel<8>, dim3((unsigned)((p->count + 7) / 8)), dim3(256), 0, stream, *p);
    else hipLaunchKernelGGL(k01_kernel<1>, dim3((unsigned)((p->count + 63) / 64)), dim3(256), 0, stream, *p);
    return hipGetLastError();
}

// ecc_launch_k01 for at most ECC_K01_WIDE_MAX_PAIRS pairs with x->patch_* (see k01_patched_kernel); nothing else of x is used.
extern "C" hipError_t ecc_launch_k01_patched(const EccPairParams* p, const EccSmallEval* x, hipStream_t stream)
{
    if (p->count <= 0) return hipSuccess;
    if (p->count > ECC_K01_WIDE_MAX_PAIRS || p->patch_count || x->patch_count < 0 || x->patch_count > ECC_SMALL_PATCH_MAX) return hipErrorInvalidValue;
    EccSmallEval xx = *x;
    xx.magic = ECC_SMALL_MAGIC;
    if (p->count <= ECC_K01_LANES16_MAX_PAIRS) {
        constexpr int per_wg = 64 / ECC_K01_SMALL_LANES;
        hipLaunchKernelGGL(k01_patched_kernel<ECC_K01_SMALL_LANES>, dim3((unsigned)((p->count + per_wg - 1) / per_wg)), dim3(256), 0, stream, *p, xx);
    } else
        hipLaunchKernelGGL(k01_patched_kernel<8>, dim3((unsigned)((p->count + 7) / 8)), dim3(256), 0, stream, *p, xx);
    return hipGetLastError();
}

// Needs the records of ecc_launch_k01 for the same parameters, earlier on the same stream.
extern "C" hipError_t ecc_launch_pairs(const EccPairParams* p, hipStream_t stream)
{
    if (p->count <= 0) return hipSuccess;
    long long nblk = (p->count + 3) / 4;
    if (p->reference_arithmetic) {
        const int stride = (p->k_limit + 63) & ~63;
        const size_t wide_lds = sizeof(float) * (size_t)stride * (p->use_corr ? 3u : 1u);
        if (p->reference_split > 1 && p->count <= ECC_REFERENCE_WIDE_MAX_PAIRS && wide_lds <= 48 * 1024) {
            // the same bits from 1024 threads per pair (see pairs_reference_wide_kernel)
            const dim3 grid((unsigned)p->count), block(ECC_REFERENCE_WIDE_THREADS);
            if (p->use_corr) hipLaunchKernelGGL((pairs_reference_wide_kernel<true>), grid, block, wide_lds, stream, *p, stride);
            else hipLaunchKernelGGL((pairs_reference_wide_kernel<false>), grid, block, wide_lds, stream, *p, stride);
        } else if (p->reference_split > 1) {
            if (p->use_corr) hipLaunchKernelGGL((pairs_reference_kernel<true, 4>), dim3((unsigned)p->count), dim3(PK_THREADS), 0, stream, *p);
            else hipLaunchKernelGGL((pairs_reference_kernel<false, 4>), dim3((unsigned)p->count), dim3(PK_THREADS), 0, stream, *p);
        } else {
            if (p->use_corr) hipLaunchKernelGGL((pairs_reference_kernel<true, 1>), dim3((unsigned)nblk), dim3(PK_THREADS), 0, stream, *p);
            else hipLaunchKernelGGL((pairs_reference_kernel<false, 1>), dim3((unsigned)nblk), dim3(PK_THREADS), 0, stream, *p);
        }
        return hipGetLastError();
    }
    // (beside_another_launch == 2 -- the moved view's pairs beside an all-pairs launch of tens of thousands: their launch gets
    // the holes retiring workgroups leave and is through long before the big one either way; as 100 workgroups of four whole
    // pairs instead of 399 of four quarter pairs it takes 30 instead of 130-180 us and the big launch 0.302 instead of 0.307 ms:
    // 3 069 / 3 068 -> 3 085 / 3 095 evaluations/s A/B/A/B on one box.  Beside a shard-sized launch the list is on the
    // critical path and keeps its four waves per pair.)
    if (!p->use_corr && !p->skip_enabled && p->beside_another_launch < 2 && p->count <= ECC_PAIRS_SPLIT_MAX) {
        // few pairs: several waves per pair (pairs_split_kernel); four while that fits one round of resident waves
        const int wpp = (p->count <= ECC_PAIRS_SPLIT8_MAX && !p->beside_another_launch) ? 8 : (p->count <= ECC_PAIRS_SPLIT4_MAX ? 4 : 2);
        const int ppw = wpp > 4 ? 1 : 4 / wpp;
        const int stride = (p->k_limit + 63) & ~63;
        const size_t lds = sizeof(float) * (size_t)ppw * (size_t)stride;
        if (lds <= 48 * 1024) {  // (a user-chosen dkappa with tens of thousands of samples per pair: the one-wave kernel)
            const dim3 grid((unsigned)((p->count + ppw - 1) / ppw)), block(wpp > 4 ? 64 * wpp : PK_THREADS);
            if (p->is_derivative) {
                if (wpp == 8) hipLaunchKernelGGL((pairs_split_kernel<true, 8>), grid, block, lds, stream, *p, stride);
                else if (wpp == 4) hipLaunchKernelGGL((pairs_split_kernel<true, 4>), grid, block, lds, stream, *p, stride);
                else hipLaunchKernelGGL((pairs_split_kernel<true, 2>), grid, block, lds, stream, *p, stride);
            } else {
                if (wpp == 8) hipLaunchKernelGGL((pairs_split_kernel<false, 8>), grid, block, lds, stream, *p, stride);
                else if (wpp == 4) hipLaunchKernelGGL((pairs_split_kernel<false, 4>), grid, block, lds, stream, *p, stride);
                else hipLaunchKernelGGL((pairs_split_kernel<false, 2>), grid, block, lds, stream, *p, stride);
            }
            return hipGetLastError();
        }
    }
    nblk = (p->count + PK_MAIN_WAVES - 1) / PK_MAIN_WAVES;
    long long per_xcd = (nblk + 7) / 8;
    dim3 grid((unsigned)(per_xcd * 8)), block(PK_MAIN_THREADS);
    if (p->use_corr) {
        if (p->is_derivative)
            hipLaunchKernelGGL((pairs_kernel<true, true>), grid, block, 0, stream, *p);
        else
            hipLaunchKernelGGL((pairs_kernel<false, true>), grid, block, 0, stream, *p);
    } else {
        if (p->is_derivative)
            hipLaunchKernelGGL((pairs_kernel<true, false>), grid, block, 0, stream, *p);
        else
            hipLaunchKernelGGL((pairs_kernel<false, false>), grid, block, 0, stream, *p);
    }
    return hipGetLastError();
}

namespace {
// The same sum over SUM_BLOCKS workgroups inside ONE launch (large counts): every workgroup reduces a contiguous slice in a
// fixed order -- one memory round trip instead of the single workgroup's two -- and publishes its float64 partial; the
// workgroup that arrives last (a ticket counter) adds the partials in slice order and stores the result.  Deterministic:
// the same slices, the same order within a slice, the same order of the partials, whichever workgroup happens to be last.
// Hand-off between workgroups as the CDNA4 guide prescribes for it (MI355X_MICROARCH.md, "Valid forms"): the partial is an
// agent-scope (sc1, write-through) store, the storing lane drains it (s_waitcnt vmcnt(0)) before its agent-scope ticket
// add, the last arriver -- told by the value its add returned -- reads the partials with agent-scope (sc1) loads.
// That hand-off is what gfx950's code generation of these operations guarantees, not what the C++ memory model does for
// relaxed atomics: the kernel is tied to the target it was written and measured for.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "sum_pairs_split_kernel's cross-workgroup hand-off is specified for gfx950 only"
#endif
constexpr int SUM_BLOCKS = 16;
struct SumScratch {
    double partial[SUM_BLOCKS];
    unsigned ticket;  // zero between launches (reset by the last arriver)
};

__global__ __launch_bounds__(1024) void sum_pairs_split_kernel(const float* __restrict__ vals, long long count,
                                                               double* __restrict__ out, SumScratch* __restrict__ scratch)
{
    __shared__ double s[1024 / 64];
    __shared__ unsigned s_ticket;
    const long long n4 = count >> 2;
    const long long per = (n4 + SUM_BLOCKS - 1) / SUM_BLOCKS;  // float4 per slice
    const long long lo = (long long)blockIdx.x * per, hi = min(n4, lo + per);
    const float4* __restrict__ v4 = reinterpret_cast<const float4*>(vals);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    long long k = lo + threadIdx.x;
    for (; k + 3 * 1024 < hi; k += 4 * 1024) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = v4[k + u * 1024];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a0 += (double)v[u].x;
            a1 += (double)v[u].y;
            a2 += (double)v[u].z;
            a3 += (double)v[u].w;
        }
    }
    for (; k < hi; k += 1024) {
        const float4 v = v4[k];
        a0 += (double)v.x;
        a1 += (double)v.y;
        a2 += (double)v.z;
        a3 += (double)v.w;
    }
    double acc = (a0 + a1) + (a2 + a3);
    if (blockIdx.x == SUM_BLOCKS - 1 && threadIdx.x == 0)
        for (long long q = n4 << 2; q < count; ++q) acc += (double)vals[q];  // the up to three values past the last float4
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double part = 0.0;
        for (int w = 0; w < 1024 / 64; w++) part += s[w];
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(&scratch->partial[blockIdx.x]),
                           (unsigned long long)__double_as_longlong(part), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_ticket = __hip_atomic_fetch_add(&scratch->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s_ticket == SUM_BLOCKS - 1) {  // every other workgroup's partial was drained before its add: all are visible
            double tot = 0.0;
            for (int b = 0; b < SUM_BLOCKS; ++b)
                tot += __longlong_as_double((long long)__hip_atomic_load(
                    reinterpret_cast<unsigned long long*>(&scratch->partial[b]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            __hip_atomic_store(&scratch->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(tot),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
}  // namespace

namespace {
// One double from device memory into a pinned, device-mapped host slot (system-scope store): how a value that a
// collective left on the device (the all-reduced sum of a sharded evaluation) reaches a polling host without a copy command.
__global__ void publish_scalar_kernel(const double* __restrict__ value, double* __restrict__ host_slot)
{
    const unsigned long long bits = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(value), __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_slot), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

extern "C" hipError_t ecc_launch_publish_scalar(const double* value_d, double* host_slot_dev, hipStream_t stream)
{
    hipLaunchKernelGGL(publish_scalar_kernel, dim3(1), dim3(1), 0, stream, value_d, host_slot_dev);
    return hipGetLastError();
}

extern "C" size_t ecc_sum_scratch_bytes() { return sizeof(SumScratch); }

// scratch: ecc_sum_scratch_bytes() of zeroed device memory owned by the caller (one per stream of launches), or null.
extern "C" hipError_t ecc_launch_sum_pairs_to_host(const float* vals, long long count, double* out, float* values_host, hipStream_t stream)
{
    hipLaunchKernelGGL(sum_pairs_kernel, dim3(1), dim3(1024), 0, stream, vals, count, out, values_host);
    return hipGetLastError();
}

extern "C" hipError_t ecc_launch_sum_pairs(const float* vals, long long count, double* out, void* scratch, hipStream_t stream)
{
    if (scratch && count >= 32768)
        hipLaunchKernelGGL(sum_pairs_split_kernel, dim3(SUM_BLOCKS), dim3(1024), 0, stream, vals, count, out,
                           static_cast<SumScratch*>(scratch));
    else
        hipLaunchKernelGGL(sum_pairs_kernel, dim3(1), dim3(1024), 0, stream, vals, count, out, (float*)nullptr);
    return hipGetLastError();
}
