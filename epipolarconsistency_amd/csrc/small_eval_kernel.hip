// small_eval_kernel.hip -- ONE launch for an evaluation of at most ECC_SMALL_EVAL_MAX_PAIRS pairs (gfx950).
//
// The reference's launcher is two kernels and two device-wide syncs per evaluation, then a host loop over the pair values
// (ref: EpipolarConsistencyRadonIntermediate.cu:300-409, ...RadonIntermediate.cpp:197-224), and its only working optimiser
// caller evaluates a handful of pairs per objective call (ref: tools/FluoroTracking/FluoroTracking.cpp:179-211).  The
// stream-ordered form of this library (e1_kernel, k01_kernel, pairs_kernel / pairs_reference_kernel, sum_pairs_kernel) costs
// such an evaluation three dependent-kernel boundaries and, for index lists, two copy commands -- more than its kernels
// take.  Here the same arithmetic runs in one kernel:
//   * E1 (per-view (P^+)^T and source position) is done by the HOST for the views whose matrix changed since the device
//     arrays were made (ecc_host_geometry.h: the code e1_kernel compiles, bit-identical) and handed over in the KERNEL
//     ARGUMENTS (up to 16 views; more: e1_kernel first); workgroup 0 stores them into the device arrays for later launches.
//     Index lists and the pair values a caller wants back travel through pinned, device-mapped memory (no copy
//     commands): the list is read once per workgroup, the values are written by the workgroup that sums.  (Round 4, first
//     form: the whole geometry in pinned memory, read by every wave over PCIe -- 30 us for 2016 pairs, 110 us for a
//     512-pair list; host memory is not cached on the device.)
//   * phase A: the pair records of the workgroup's pairs by k01_fit_block<8> -- the code of k01_kernel<8>, 8 lanes per fit
//     -- into LDS;
//   * phase B: the sampling loops of pairs_kernel (pair_accumulate) or of pairs_reference_kernel<.., 4> (reference_loop).
//     With few pairs most SIMDs would idle, so the four waves of a workgroup share ONE pair (WPP = 4): wave
//     `sub` takes the 64-sample trips sub, sub + WPP, ... and stores every sample's term in LDS; the pair's first wave
//     then adds the terms per lane in the order ONE wave accumulates them (k = lane, lane + 64, ...), so the pair value has
//     the bits of pairs_kernel's.  The reference arithmetic keeps its own grouping (thread T: k = T, T + 256, ...; wave
//     sums in wave order), which is what pairs_reference_kernel<.., 4> does for the same evaluation sizes -- from 1024
//     threads per pair (pairs_reference_wide_kernel's scheme: terms staged in LDS, added by the first 256 threads in that
//     order; the records are made by the first four waves);
//   * phase C: each value goes to the device array (plain store) and, at system scope, into pinned host memory; the
//     workgroup drains its stores and takes a ticket, and the workgroup that arrives last writes a "done" word the host
//     polls.  The HOST then adds the values in sum_pairs_kernel's order (the float4 layout of its 1024 threads, its shuffle
//     tree, its 16 wave sums in order: ecc_evaluate.hip, small_sum_on_host) -- at most 4096 values, under 2 us.  (First form,
//     measured: the last arriver added the values itself with agent-scope loads -- 55 us for 399 values; loads that must
//     bypass the XCD's L2 cost microseconds each.  Asynchronous callers, who want the sum in device memory, keep the
//     stream-ordered launches.)
// Every result is bit-identical to the multi-launch path (tests/test_gpu_small_eval.py); ecc_metric_set_small_eval(0)
// keeps the old path.
#include <hip/hip_runtime.h>
#include <float.h>
#include <cstdlib>

#include "ecc_layout.h"
#include "ecc_pairs_device.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "small_eval_kernel's cross-workgroup hand-off is specified for gfx950 only"
#endif

namespace {

__device__ __forceinline__ void store_system(float* dst, float v)
{
    __hip_atomic_store(reinterpret_cast<unsigned*>(dst), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// WPP: waves per pair (4, 2, 1); REF: ECC_SAMPLING_REFERENCE (one pair per workgroup of 1024 threads, WPP = 4 names the
// grouping of pairs_reference_kernel<.., 4> its sums reproduce).
#ifndef ECC_SMALL_MIN_WAVES
#define ECC_SMALL_MIN_WAVES 4
#endif
template <bool DERIV, int WPP, bool REF>
__global__ __launch_bounds__(REF ? 1024 : 256, ECC_SMALL_MIN_WAVES) void small_eval_kernel(EccPairParams p, EccSmallEval x)
{
    constexpr int PPW = 4 / WPP;  // pairs per workgroup
    static_assert(!REF || WPP == 4, "reference arithmetic: one pair per workgroup");
    static_assert(PPW <= 64 / ECC_K01_SMALL_LANES, "phase A makes 64 / ECC_K01_SMALL_LANES records per workgroup");
    extern __shared__ float stage_all[];  // WPP > 1 or REF: PPW * x.stage_stride floats
    const bool member = !REF || threadIdx.x < 256;  // wave-uniform: the threads that make the records
    __shared__ K01Shared<ECC_K01_SMALL_LANES> ks;
    __shared__ int32_t idx_lds[4 * PPW];
    __shared__ double part[4];

#ifdef ECC_SMALL_STAMPS  // experiments (scripts/exp_small_phases.py); the stamps cost registers: not in the product build
#define ECC_SMALL_STAMP(i) do { if (x.dbg && threadIdx.x == 0) x.dbg[4 * blockIdx.x + (i)] = wall_clock64(); } while (0)
#else
#define ECC_SMALL_STAMP(i)
#endif
    ECC_SMALL_STAMP(0);

    // ---- phase A: records of this workgroup's pairs (k01_kernel<8>'s code; dead slots take part in its exchanges) ----
    const long long blk_first = (long long)blockIdx.x * PPW;
    if (p.indices) {  // uniform over the launch: the tuples of this workgroup's pairs, one read of the pinned list
        if (threadIdx.x < 4 * PPW && blk_first + (threadIdx.x >> 2) < p.count) idx_lds[threadIdx.x] = p.indices[4 * blk_first + threadIdx.x];
        __syncthreads();
    }
    // the patch list is read in place, from the kernel-argument segment (explicit arguments start at offset 0, `x` follows
    // `p` at its 8-byte alignment); magic: the two views of the same argument must agree, or the launch reports NaN
    typedef const char __attribute__((address_space(4))) * KernargBytes;
    static_assert(alignof(EccSmallEval) == 8 && alignof(EccPairParams) == 8, "layout of the kernel arguments");
    const EccSmallEvalArg xs = (EccSmallEvalArg)((KernargBytes)__builtin_amdgcn_kernarg_segment_ptr() + ((sizeof(EccPairParams) + 7) & ~(size_t)7));
    const bool args_ok = xs->magic == ECC_SMALL_MAGIC && x.magic == ECC_SMALL_MAGIC && xs->patch_count == x.patch_count;
    k01_fit_block<ECC_K01_SMALL_LANES>(p, blk_first, PPW, ks, xs, p.indices ? idx_lds : nullptr, member);  // ends with a barrier

    ECC_SMALL_STAMP(1);
    // ---- phase B ----
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = REF ? 0 : wave / WPP, sub = REF ? wave : wave % WPP;  // wave-uniform
    const long long local = blk_first + slot;
    const bool live = local < p.count;
    const EccPairRecord* rec = &ks.recs[slot];
    float val = 0.f;
    if (REF) {
        double acc = 0.0, m2 = 0.0, m3 = 0.0, m4 = 0.0;
        if (live) {
            float K0[8], K1[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                K0[i] = uniformf(rec->K0[i]);
                K1[i] = uniformf(rec->K1[i]);
            }
            const int iD0 = __builtin_amdgcn_readfirstlane(rec->iD0), iD1 = __builtin_amdgcn_readfirstlane(rec->iD1);
            reference_loop<false, true>(p, K0, K1, (GlobalFloats)p.slabs[iD0], (GlobalFloats)p.slabs[iD1], (int)threadIdx.x, 1024, acc,
                                        m2, m3, m4, stage_all, x.stage_stride);
        }
        __syncthreads();  // every sample's term is staged
        if (wave < 4) {   // the sums of a 256-thread workgroup
            if (live) reference_resum<false>(p, uniformf(rec->K1[6]), uniformf(rec->K1[7]), (int)threadIdx.x, stage_all, x.stage_stride, acc, m2, m3, m4);
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
            if (lane == 0) part[wave] = acc;
        }
        __syncthreads();
        acc = part[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) acc += part[w];  // wave sums in wave order, as pairs_reference_kernel<.., 4>
        val = (float)acc;
    } else {
        double acc = 0.0, m2 = 0.0, m3 = 0.0, m4 = 0.0;
        float* stage = stage_all + (size_t)slot * x.stage_stride;
        int iD0 = 0, iD1 = 0;
        if (live) {
            iD0 = __builtin_amdgcn_readfirstlane(rec->iD0);
            iD1 = __builtin_amdgcn_readfirstlane(rec->iD1);
            pair_accumulate<DERIV, false, WPP>(p, rec, iD0, iD1, lane, acc, m2, m3, m4, sub, stage);
        }
        if (WPP > 1) {
            __syncthreads();  // every wave of the pair has stored its trips
            if (live && sub == 0) {
                // the terms in the order one wave accumulates them: lane l adds k = l, l + 64, ... until kappa reaches kappa_max
                const float dkappa = uniformf(rec->K1[6]), kappa_max = uniformf(rec->K1[7]);
                for (int k = lane; k < p.k_limit; k += 64) {
                    const float kappa = dkappa * 0.5f + dkappa * k;  // ref: ...RadonIntermediate.cu:259 (same fp32 ops)
                    if (kappa >= kappa_max) break;
                    acc += (double)stage[k];
                }
            }
        }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
        val = (float)acc;
    }

    // ---- phase C: the pair values; the workgroup that arrives last announces them ----
    if (live && sub == 0 && lane == 0) {
        if (p.pair_values) p.pair_values[local] = val;  // the device copy (later list launches, copies to the caller): visible at kernel end
        if (p.cost && !p.indices) {
            const int ci = rec->ci, cj = rec->cj;
            p.cost[(size_t)ci + (size_t)cj * p.n_views] = val;
        }
        if (x.values_host) {
            store_system(x.values_host + local, val);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // on its way to the host before this workgroup's ticket
        }
    }
    if (!x.done_out) return;  // uniform over the launch
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(x.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == gridDim.x - 1) {
            // every other workgroup's values were drained before its ticket add, and writes of one device to host memory
            // arrive in order: the host that sees this word sees all values
            __hip_atomic_store(x.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero between launches
            __hip_atomic_store(x.done_out, args_ok ? x.done_token : ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace

// Host-side plan of the launch: waves per pair by evaluation size, LDS for the staged terms.  Returns 0 when the
// evaluation cannot take this path (the caller falls back to the stream-ordered launches).
extern "C" int ecc_small_eval_plan(const EccPairParams* p, long long forced_bound, int* wpp, size_t* lds_bytes)
{
    // forced_bound >= 0: ecc_debug_set_small_eval_bound (experiments)
    const long long bound = forced_bound >= 0 ? forced_bound : ECC_SMALL_EVAL_PAIR_BOUND(p->k_limit);
    if (p->count < 1 || p->count > bound || p->count > ECC_SMALL_EVAL_MAX_PAIRS || p->use_corr || p->K01_out || p->record_slots || p->value_slots ||
        p->patch_count || p->skip_enabled)
        return 0;
    if (p->reference_arithmetic) {
        if (p->reference_split != 4) return 0;
        const size_t ref_bytes = sizeof(float) * (size_t)((p->k_limit + 63) & ~63);
        if (ref_bytes > 40 * 1024) return 0;
        *wpp = 4;
        *lds_bytes = ref_bytes;
        return 1;
    }
    const int w = 4;  // waves per pair (the kernel has forms with 2 and 1 for larger launches: measured slower than the stream-ordered path there)
    const size_t bytes = sizeof(float) * (size_t)((p->k_limit + 63) & ~63);
    if (bytes > 40 * 1024) return 0;  // user-chosen dkappa with tens of thousands of samples per pair
    *wpp = w;
    *lds_bytes = bytes;
    return 1;
}

extern "C" hipError_t ecc_launch_small_eval(const EccPairParams* p, const EccSmallEval* x, hipStream_t stream)
{
    int wpp = 0;
    size_t lds = 0;
    if (!ecc_small_eval_plan(p, p->count, &wpp, &lds)) return hipErrorInvalidValue;  // (the caller has applied its size bound)
    EccSmallEval xx = *x;
    xx.stage_stride = (p->k_limit + 63) & ~63;
    xx.magic = ECC_SMALL_MAGIC;
    const unsigned blocks = (unsigned)((p->count + (4 / wpp) - 1) / (4 / wpp));
    const dim3 grid(blocks), block(p->reference_arithmetic ? 1024 : 256);
#define ECC_SMALL(D, W, R) hipLaunchKernelGGL((small_eval_kernel<D, W, R>), grid, block, lds, stream, *p, xx)
    if (p->reference_arithmetic) ECC_SMALL(true, 4, true);  // (the reference arithmetic takes is_derivative at run time)
    else if (p->is_derivative) ECC_SMALL(true, 4, false);
    else ECC_SMALL(false, 4, false);
#undef ECC_SMALL
    return hipGetLastError();
}
