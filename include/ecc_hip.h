/*
 * ecc_hip.h -- C ABI of the MI355X-native epipolar-consistency hot path (libecc_hip.so).
 *
 * Drop-in boundary for the reference's Radon-intermediate + pair-consistency path
 * (aaichert/EpipolarConsistency 1.2.2).  Plain C: opaque handles, raw pointers, int32 sizes,
 * every function returns an int status (0 = ECC_OK), no exceptions, no exit().  Citations are
 * relative to /root/reference/code/.  The reference-side bindings (what a maintainer adds to
 * RadonIntermediate.cpp / EpipolarConsistencyRadonIntermediate.cpp) are shown in INTEGRATION.md;
 * a header-only C++ adapter with the reference's class names lives in
 * epipolarconsistency_amd/cpp/EpipolarConsistencyHip.hxx.
 *
 * Layout contracts (same as the reference unless stated):
 *   - projection images: row-major float32, x (u) fastest, n_u x n_v;
 *   - Radon intermediate ("dtr") as seen through this API: n_t rows x n_alpha columns float32,
 *     alpha fastest, idx = iy*n_alpha + ix            (ref: LibEpipolarConsistency/RadonIntermediate.cu:44);
 *   - projection matrices: 3x4 float64 column-major, 12 doubles per view (Eigen default; the
 *     reference passes Ps[i].data(), ref: ...RadonIntermediate.cpp:148);
 *   - cost image: n x n float32, entry (i,j), i<j at index i + j*n; other entries untouched
 *     (ref: EpipolarConsistencyRadonIntermediate.cu:250,269; .cpp:183,214-221);
 *   - index lists: int32 x 4 per pair = (P0, P1, dtr0, dtr1)    (ref: ...RadonIntermediate.cu:49-50,202).
 * Device-resident dtrs are kept in a private transposed+padded layout (ECC_LAYOUT_T_FAST, see
 * DESIGN.md "Data layout in HBM"); ecc_dtr_readback/ecc_dtr_from_host convert.
 */
#ifndef ECC_HIP_H
#define ECC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ECC_HIP_VERSION 100 /* 1.0.0 */

/* status codes */
enum {
    ECC_OK = 0,
    ECC_ERR_INVALID_ARGUMENT = 1,
    ECC_ERR_HIP = 2,          /* a HIP runtime call failed: see ecc_last_error() */
    ECC_ERR_OUT_OF_MEMORY = 3,
    ECC_ERR_NO_DEVICE = 4,
    ECC_ERR_UNSUPPORTED = 5
};

/* RadonIntermediate::Filter / ::PostProcess (ref: LibEpipolarConsistency/RadonIntermediate.h:21-29) */
enum { ECC_FILTER_DERIVATIVE = 0, ECC_FILTER_RAMP = 1, ECC_FILTER_NONE = 2 };
enum { ECC_POST_IDENTITY = 0, ECC_POST_SQUARE_ROOT = 1, ECC_POST_LOGARITHM = 2 };

typedef struct ecc_ctx ecc_ctx;       /* one per (device, stream) */
typedef struct ecc_dtr ecc_dtr;       /* one Radon intermediate  (ref: class RadonIntermediate) */
typedef struct ecc_metric ecc_metric; /* ref: class MetricRadonIntermediate */

/* Thread-local message of the last failing call on this thread ("" if none). */
const char* ecc_last_error(void);
int ecc_version(void);
/* Number of HIP devices visible (0 when there is none; never fails). */
int ecc_device_count(void);

/* Environment variables the library reads -- two, both optional, neither changes a result; each is described next to
 * the entry point it affects:
 *   ECC_RECORD_REUSE = 0 | 1 | 2   default of ecc_metric_set_record_reuse for new metrics   (see ecc_metric_set_record_reuse)
 *   ECC_EXCHANGE_TIMEOUT_S         time-out of ecc_exchange_sum in seconds, default 60      (see ecc_exchange_open)
 * (the header-only C++ adapter additionally reads ECC_HIP_DEVICES: the devices of its process-wide default group).
 * Experiment hooks are explicit calls (ecc_debug_*, last section of this header), never environment variables: the
 * variables of earlier versions -- ECC_POLY_TOL, ECC_SMALL_MAX_PAIRS, ECC_SMALL_DEBUG, ECC_RESULT_WAIT, ECC_QUAD_COPIES --
 * are ignored (tests/test_gpu_env_hooks.py). */

/* ---- context ------------------------------------------------------------------------------ */
/* `stream` is a hipStream_t (may be NULL = the device's default stream).  All launches and
 * copies of objects created from this context are ordered on it.  Replaces the reference's
 * implicit "current device, default stream, cudaDeviceSynchronize after every launch". */
int ecc_ctx_create(int device, void* stream, ecc_ctx** out);
/* Destroy every dtr / metric created from a context BEFORE the context itself (they keep a pointer to it). */
int ecc_ctx_destroy(ecc_ctx* ctx);
int ecc_ctx_synchronize(ecc_ctx* ctx);

/* ---- Radon intermediate (R1/R2) ------------------------------------------------------------ */
/* Replaces computeDerivLineIntegrals(tex, n_x, n_y, n_alpha, n_t, filter, post, out_d)
 * (ref: LibEpipolarConsistency/RadonIntermediate.cu:149-170) and the ctor
 * RadonIntermediate(ImageView<float>, size_alpha, size_t, filter, post) (ref: RadonIntermediate.cpp:17-31).
 * `image` is n_u*n_v floats on the host (image_on_device = 0) or on ctx's device (= 1).
 * ECC_FILTER_RAMP: plain line integrals followed by the ramp filter along t of apply1DRampFilter
 * (ref: RadonIntermediate.cu:173-237), evaluated as the exact circular convolution the two cuFFT
 * calls amount to (binary64 accumulation, no FFT library). */
int ecc_radon_compute(ecc_ctx* ctx, const float* image, int image_on_device, int n_u, int n_v,
                      int n_alpha, int n_t, int filter, int post_process, ecc_dtr** out);

/* Same for a stack of n images (n*n_u*n_v floats); writes n handles to out[0..n).  The n dtrs
 * share one device slab; each handle is destroyed separately.  Work is asynchronous on the
 * context's stream. */
int ecc_radon_compute_batch(ecc_ctx* ctx, const float* images, int images_on_device, int n, int n_u,
                            int n_v, int n_alpha, int n_t, int filter, int post_process,
                            ecc_dtr** out);

/* Device-to-device form for callers that own both buffers (e.g. torch tensors): images_d holds n
 * images, slabs_d receives n dtrs in the private layout (ecc_dtr_slab_floats(n_alpha, n_t) floats
 * each, written completely incl. the replicated border).  Asynchronous on the context's stream;
 * adopt the result with ecc_dtr_wrap_device.  A metric that already holds handles on these slabs keeps
 * sampling its snapshot of the old contents until ecc_metric_refresh_dtrs (see there). */
int ecc_radon_compute_into(ecc_ctx* ctx, const float* images_d, int n, int n_u, int n_v, int n_alpha,
                           int n_t, int filter, int post_process, float* slabs_d);

/* Arithmetic convention of the Radon kernel's sampling loop (per context; default ECC_RADON_EXACT).
 *   ECC_RADON_EXACT: every float expression of ref: RadonIntermediate.cu:118-123 and of the bilinear rule
 *     ((1 - fx) * T00 + fx * T10, ...) rounded separately, source order -- the CPU reading of the source; results are
 *     bit-identical to the oracle's normative variant.
 *   ECC_RADON_FMA: the loop body contracted -- positions fmaf(t, d, o), lerps as T00 + fx * (T10 - T00) with one fused
 *     multiply-add each (3 differences + 3 fma instead of 11 operations per sample).  This is the arithmetic class of
 *     the reference's own GPU build, which interpolates in texture hardware (ref: LibUtilsCuda/CudaBindlessTexture.cpp:
 *     25-39) and whose compiler contracts o + t * d; results are bit-identical to the oracle's contracted variant
 *     (eccor_set_radon_contract(1)) and move the ECC metric by less than 2e-6 relative (tests/test_gpu_radon_fma.py,
 *     DESIGN.md 4.1).  About 25 % faster.
 * The per-bin set-up (line, clipping, bounds test), the accumulation order and the post-process are the same in both. */
enum { ECC_RADON_EXACT = 0, ECC_RADON_FMA = 1 };
int ecc_radon_set_arithmetic(ecc_ctx* ctx, int mode);
int ecc_radon_get_arithmetic(const ecc_ctx* ctx, int* mode);

/* The reference's launcher seam, device memory in and out (Option B): replaces
 *   void computeDerivLineIntegrals(cudaTextureObject_t in, int n_x, int n_y, int n_alpha, int n_t, int filter, int post, float* out_d)
 * (ref: RadonIntermediate.cpp:12, RadonIntermediate.cu:149-170) with the image as a linear device buffer in place of the
 * texture.  out_linear_d receives the result in the REFERENCE's layout: n_t rows of n_alpha floats, angle fastest,
 * n_t * n_alpha floats -- the buffer RadonIntermediate::compute allocates (ref: RadonIntermediate.cpp:208) and readback
 * copies verbatim (:148-163).  Stream-ordered. */
int ecc_radon_compute_linear(ecc_ctx* ctx, const float* image_d, int n_u, int n_v, int n_alpha, int n_t, int filter,
                             int post_process, float* out_linear_d);
/* A Radon intermediate from device memory in the reference's layout: what RadonIntermediate::getTexture turns into a
 * texture (ref: RadonIntermediate.cpp:188-196) -- a snapshot, as there. */
int ecc_dtr_from_device_linear(ecc_ctx* ctx, const float* data_d, int n_alpha, int n_t, int n_u, int n_v, int filter,
                               ecc_dtr** out);

/* Wraps existing host data (alpha-fast, n_t x n_alpha), ref: RadonIntermediate(ImageView<float>)
 * + replaceRadonIntermediateData (RadonIntermediate.cpp:69-80,105-123). */
int ecc_dtr_from_host(ecc_ctx* ctx, const float* data, int n_alpha, int n_t, int n_u, int n_v,
                      int filter, ecc_dtr** out);
/* ref: RadonIntermediate::readback (RadonIntermediate.cpp:148-163); host gets alpha-fast data. */
int ecc_dtr_readback(ecc_dtr* dtr, float* host_out);
/* Getters, ref: RadonIntermediate.cpp:130-133,165-178.  bin_size_angle = Pi/n_alpha,
 * bin_size_distance = sqrt(n_u^2+n_v^2)/n_t (RadonIntermediate.cpp:204-206). */
int ecc_dtr_info(const ecc_dtr* dtr, int* n_alpha, int* n_t, int* n_u, int* n_v, int* filter,
                 double* bin_size_angle, double* bin_size_distance);
/* Device pointer + pitch of the private layout: element (ix, iy) lives at
 * base[(ix + 1) * pitch + (iy + 1)], rows -1 and n_alpha and columns -1 and n_t replicate the
 * border (clamp addressing).  For callers that want to all-gather dtrs between GPUs. */
int ecc_dtr_device_view(const ecc_dtr* dtr, float** base, int* pitch, int* rows);
/* Size in floats of one dtr in the private layout, so that a caller (e.g. torch) can own the slab. */
int64_t ecc_dtr_slab_floats(int n_alpha, int n_t);
/* Adopt caller-owned device memory already holding a dtr in the private layout (no copy, no free).  The handle
 * aliases the caller's memory: see ecc_metric_refresh_dtrs for what a metric sees when the memory changes. */
int ecc_dtr_wrap_device(ecc_ctx* ctx, float* base, int n_alpha, int n_t, int n_u, int n_v, int filter,
                        ecc_dtr** out);
int ecc_dtr_destroy(ecc_dtr* dtr);

/* ---- projection pre-processing (the step in front of R1) ------------------------------------- */
/* ref: struct EpipolarConsistency::PreProccess (Gui/PreProccess.h:14-50), same fields and defaults
 * (ecc_preprocess_defaults).  Offsets in zero/feather are left, right, bottom, top; blanks are
 * n_blanks x 4 int32 (x0, y0, x1, y1) on the host. */
typedef struct ecc_preprocess_config {
    int32_t process;            /* 1: run PreProccess::process; 0: cosine weighting only            */
    int32_t normalize;          /* Intensity/Normalize                                               */
    double bias, scale;         /* Intensity/Bias, Intensity/Scale                                   */
    int32_t apply_log;          /* Intensity/Apply Minus Logarithm                                   */
    double gaussian_sigma;      /* Lowpass Filter/Gaussian Sigma   (1.84)                            */
    int32_t half_kernel_width;  /* Lowpass Filter/Half Kernel Width (5); at most 16 here            */
    int32_t flip_u, flip_v;     /* Geometry/Flip u-Axis, Flip v-Axis                                 */
    int32_t zero[4];            /* Border/Zero Border (1,1,1,1)                                      */
    int32_t feather[4];         /* Border/Feather (16,16,16,16)                                      */
    int32_t n_blanks;
    const int32_t* blanks;      /* Border/Blanks                                                     */
} ecc_preprocess_config;
void ecc_preprocess_defaults(ecc_preprocess_config* cfg);

/* ref: PreProccess::process(img) followed by PreProccess::apply_weight_cos_principal_ray(img, P)
 * (Gui/PreProccess.cpp:57-166; call order of Gui/InputDataDirect.cpp:85-86) for a stack of n images, as ONE
 * fused device kernel (intensity, border zero/feather, blanks, flips, Gaussian low-pass with the reference's
 * dropped last tap, cosine weight).  images / out: n * n_u * n_v floats, both on the host
 * (on_device = 0) or both on ctx's device (= 1); out may equal images (in place, like the reference).
 * Ps: n x 12 float64 column-major or NULL (no cosine weighting; an all-zero matrix skips that view,
 * PreProccess.cpp:149).  on_device = 1: asynchronous on the context's stream, in place (through a scratch stack kept
 * in the context) or out of place; `out` must be identical to `images` or disjoint from it (partial overlap is
 * rejected: tiles read halos of their neighbours).  on_device = 0 returns with `out` filled. */
int ecc_preprocess(ecc_ctx* ctx, const float* images, int on_device, float* out, int n, int n_u, int n_v,
                   const ecc_preprocess_config* cfg, const double* Ps);
/* K(0,0), K(0,2), K(1,2) of P = K[R|t] (ref: getCameraIntrinsics, ProjectionMatrix.cpp:61-67). */
void ecc_host_intrinsics(const double* P, float* sdd_px, float* ppu, float* ppv);

/* ---- metric (E1..E5) ----------------------------------------------------------------------- */
/* ref: MetricRadonIntermediate(Ps, dtrs) / setRadonIntermediates (…RadonIntermediate.cpp:53-66,87-106).
 * The metric borrows the dtrs ("DO NOT delete or change _dtrs during lifetime", .h:45).  Sizes and
 * the derivative flag are taken from dtrs[0] only, as the reference does (.cpp:92-98). */
int ecc_metric_create(ecc_ctx* ctx, int n_dtrs, ecc_dtr* const* dtrs, ecc_metric** out);
int ecc_metric_destroy(ecc_metric* m);
/* SNAPSHOT SEMANTICS.  ecc_metric_create copies every dtr into a private row-paired layout and the
 * all-pairs / range / index-list evaluations in the polynomial and per-sample modes sample THOSE copies: new slab
 * contents (ecc_radon_compute_into into the same slabs, an image-domain correction loop) are not seen until
 * ecc_metric_refresh_dtrs(m, first, count) re-copies dtrs [first, first + count) -- asynchronous on the context's
 * stream, ~1 us per dtr, ordered behind the kernels that wrote the slabs when they ran on the same stream.
 * ECC_SAMPLING_REFERENCE and ecc_metric_evaluate_for_image_pair read the slabs themselves (always current).  The
 * handles must stay alive for the metric's lifetime either way ("DO NOT delete or change _dtrs", ref: .h:45). */
int ecc_metric_refresh_dtrs(ecc_metric* m, int first, int count);

/* ref: MetricRadonIntermediate::setProjectionMatrices (…RadonIntermediate.cpp:134-163): per view
 * (P^+)^T and the source position in float64 (same Householder-QR arithmetic as
 * culaut/xprojectionmatrix.hxx:20-52,93-105), cast to float32.  The n x 12 doubles are uploaded
 * asynchronously and the pre-compute runs on the device, one thread per view (bit-identical to
 * ecc_host_pinvT / ecc_host_source_position). */
int ecc_metric_set_projections(ecc_metric* m, const double* Ps, int n_views);
/* The reference's launcher seam for the metric (Option B): what
 *   void epipolarConsistency(int n_x, int n_y, int num_dtrs, char* dtrs_d, int n_alpha, int n_t, float step_alpha, float step_t,
 *                            int num_Ps, float* Cs_d, float* PinvTs_d, int num_pairs, int* indices_d, float* K01s_d, float* out_d,
 *                            float object_radius_mm, float dkappa, bool isDerivative, bool use_corr, float* out_corr_d)
 * (ref: EpipolarConsistencyRadonIntermediate.cpp:16-37, .cu:300-409) does with the caller's device buffers: Cs_d / PinvTs_d
 * are the per-view geometry the caller's host class made (ref: ...RadonIntermediate.cpp:134-163), indices_d the optional
 * int4 list, K01s_d (nullable) the 16 floats per pair, out_d the n x n cost image (all pairs: entry i + j n, i < j; the rest
 * untouched) or num_pairs values (index list).  The Radon intermediates are the metric's (ecc_dtr_from_device_linear +
 * ecc_metric_create, once per data set: the reference builds its textures once too).  Synchronous, like the reference's
 * launcher.  With use_corr the value is the finished cost 1 - cc (the reference hands five moments to its host epilogue,
 * ref: .cu:116-149, .cpp:199-210: that epilogue goes).  The rest of the caller's host epilogue stays (ref: .cpp:197-224). */
int ecc_metric_evaluate_external(ecc_metric* m, int num_Ps, const float* Cs_d, const float* PinvTs_d, int num_pairs,
                                 const int32_t* indices_d, float* K01s_d, float* out_d, float object_radius_mm, float dkappa,
                                 int use_corr);
/* Debug: the device-side result of the pre-compute (12 + 4 floats per view, host output). */
int ecc_metric_debug_geometry(ecc_metric* m, float* PinvTs, float* Cs);

/* ref: Metric::setObjectRadius / setEpipolarPlaneStep / MetricRadonIntermediate::useCorrelation
 * (EpipolarConsistency.cpp:70-90, …RadonIntermediate.cpp:80-85).  0 = automatic for both scalars.
 * use_corr != 0: pair value = 1 - un-centred correlation of the two redundant signals, with the
 * reference's provisional per-sample weight kappa_max/kappa (ref: ...RadonIntermediate.cu:116-149,274;
 * .cpp:127-131); the reference has no test for it (parity unpinned). */
int ecc_metric_set_params(ecc_metric* m, double object_radius_mm, double dkappa, int use_corr);
/* How the pair kernel obtains the sample positions of a pair (not in the reference, which has one GPU path with
 * hardware texture filtering and no CPU twin).  All modes implement the same formulas; they differ in
 * fp32 rounding, which matters for a SINGLE pair: its value is a sum of squared, nearly cancelling differences and
 * moves by ~2e-5 (median) when sample positions move by 1e-5 bins, while means over many pairs average that out.
 *   ECC_SAMPLING_POLYNOMIAL  sample coordinates from per-pair polynomials in kappa fitted by the pair-geometry kernel
 *                            (pairs whose fit is rejected take PER_SAMPLE by themselves).  Fastest;
 *                            all-pairs means agree with the CPU path to < 1e-6, single pairs to ~1e-4.
 *   ECC_SAMPLING_PER_SAMPLE  every pair evaluates line -> lineToSampleDtr per sample with device approximations
 *                            (v_rsq, v_rcp + minimax atan); ~1.5x the kernel time, same accuracy class as POLYNOMIAL.
 *   ECC_SAMPLING_REFERENCE   the CPU path's arithmetic operation for operation (ref: EpipolarConsistencyCommon.hxx:152-171,
 *                            ...RadonIntermediate.cu:71-113 in fp32 source order, sin/cos/atan2 correctly rounded,
 *                            exact fp32 bilinear rule with index clamps): single pair values agree with the CPU path
 *                            to float rounding of the final sum.  ~10x the kernel time of POLYNOMIAL per pair; ranges
 *                            of at most 2048 pairs put four waves on every pair, so a small evaluation takes as long
 *                            as in the other modes (~46 us for one pair).
 *   ECC_SAMPLING_AUTO        (default) REFERENCE when one evaluation covers at most
 *                            ECC_SAMPLING_AUTO_REFERENCE_PAIRS pairs (the whole evaluation is one pair's latency either
 *                            way: 2-view metric values, index lists as in tools/Registration/Registration3D3D.hxx:95,109),
 *                            POLYNOMIAL above.  Callers that compare values ACROSS calls of different size fix the mode.
 * WHOSE ROUNDING a cost-image entry carries (measured, scripts/pair_error_attribution.py, tests/test_configs.py; bench.py reports
 * it at full size as pair_rel_err_attribution).  The same formulas with the line -> (angle, distance) mapping in binary64, rounded
 * once, give the noise-free pair values.  The REFERENCE arithmetic (fp32 in source order = ECC_SAMPLING_REFERENCE = the CPU path)
 * is p50 1.4e-5 / p99 7.0e-5 / max 1.3e-4 away from them at 64 views of 512^2 -- and its MEAN 9.9e-6, the systematic part being
 * its float constant Pi (EpipolarConsistencyCommon.hxx:155,159).  POLYNOMIAL is 9.8e-6 / 5.0e-5 / 1.2e-4 away from the same
 * noise-free values, PER_SAMPLE 9.9e-6 / 3.9e-5 / 5.8e-5: both are CLOSER to them than the reference arithmetic is, while they
 * reproduce the reference's mean (float Pi included) to 6e-8 / 1.6e-7.  So the ~1e-5 (median) by which a single entry of the
 * n x n cost image differs from the CPU path's is the fp32 rounding of the reference's own arithmetic, not an error this
 * library adds; a caller that needs the CPU path's entry bit for bit selects ECC_SAMPLING_REFERENCE. */
/* ECC_SAMPLING_AUTO resolves from the size of the EVALUATION, not of the launch: n (n - 1) / 2 for ecc_metric_evaluate_all
 * and for every ecc_metric_evaluate_range[_async] shard of it (so G shard sums add up to the one-device sum's arithmetic
 * whatever G, and cost-balanced or re-balanced shards never mix modes), the list length for index lists. */
enum { ECC_SAMPLING_AUTO = 0, ECC_SAMPLING_POLYNOMIAL = 1, ECC_SAMPLING_PER_SAMPLE = 2, ECC_SAMPLING_REFERENCE = 3 };
#define ECC_SAMPLING_AUTO_REFERENCE_PAIRS 512
int ecc_metric_set_sampling(ecc_metric* m, int mode);
/* Opt-in pose-delta evaluation (default off; not in the reference).  The reference's optimisation problems change ONE
 * view's matrix per cost-function call and call setProjectionMatrices + evaluate() over all pairs
 * (ref: Gui/SingleImageMotion.h:84-90, Gui/Visualization.h:78-98).  When enabled, ecc_metric_evaluate_all (without a cost
 * image) and ecc_metric_evaluate_range keep the pair values of their last evaluation on the device; the next call with
 * the same range and parameters compares the new matrices with the ones those values belong to and re-evaluates only
 * the pairs that contain a changed view (when at most a quarter of the views changed, otherwise everything), then sums
 * the whole array again.  The result is BIT-IDENTICAL to a full evaluation: a pair's value is a function of its two
 * matrices, its two Radon intermediates and the parameters only, the sampling mode is the one the full range resolves
 * to, and the float64 sum has a fixed order.  Changing parameters, sampling mode, the range, the number of views or
 * calling ecc_metric_refresh_dtrs drops the kept values.  ecc_metric_last_evaluated_pairs reports how many pairs the
 * last evaluate_all / evaluate_range recomputed. */
int ecc_metric_set_incremental(ecc_metric* m, int enable);
int ecc_metric_last_evaluated_pairs(const ecc_metric* m, int64_t* pairs);
/* Record reuse (default ON; not in the reference; ECC_RECORD_REUSE=0 in the environment turns the default off).  The
 * per-pair geometry of an evaluation (the reference's K01 array, ref: ...RadonIntermediate.cu:13-67, here k01_kernel's
 * records) is a function of the pair's two matrices and the parameters only.  ecc_metric_evaluate_all /
 * ecc_metric_evaluate_range[_async] keep the records of their last evaluation; when the next call evaluates the same range
 * with the same parameters and at most a quarter of the matrices differ bit-wise from the ones the records were made
 * from, only the pairs that contain a changed view are refitted, E1 (ref: ...RadonIntermediate.cpp:134-163) of the
 * changed views is computed on the host by the same code, and e1_kernel is not launched.  EVERY pair is still sampled by
 * the pair kernel: the work that is skipped is geometry that would be recomputed to identical bits, so results are
 * bit-identical with the switch on or off (tests/test_gpu_record_reuse.py).  This is the callers' pattern: the reference's
 * optimisation problems overwrite one view's matrix per cost-function call (ref: Gui/SingleImageMotion.h:84-90).
 * With the switch on, ecc_metric_set_projections only stages the matrices; E1 runs with the next call that needs it.
 * The refit of the changed pairs and their own pair-kernel launch run on a stream of the metric's own beside the
 * all-pairs launch (which skips them) on the context's stream; an event orders that stream after whatever this metric had
 * queued on the context's stream -- it reads nothing anybody else writes: records, device geometry, the row-paired copies
 * (ecc_metric_refresh_dtrs), pinned lists -- and is left out when the previous call was a synchronous evaluation, which
 * returns only after it has seen its result; the final sum is ordered after both streams (up to 512 views, no cost image;
 * otherwise one stream). */
#define ECC_RECORD_REUSE_MIN_PAIRS 4096
#ifndef ECC_RECORD_REUSE_SPLIT_PAIRS
#define ECC_RECORD_REUSE_SPLIT_PAIRS 8192
#endif
/* on: 0 = off; 1 (default) = by size: ranges of more than ECC_RECORD_REUSE_MIN_PAIRS pairs (up to there refitting
 * everything is one short launch), the two-stream form from ECC_RECORD_REUSE_SPLIT_PAIRS pairs on (a shorter pair kernel
 * cannot hide the refit's launches: measured 2016 pairs, 61 us per step with two streams, 40 us refitting everything; a
 * 10 873-pair shard 63 us with, 77 us without); 2 = the two-stream form for every size (tests). */
int ecc_metric_set_record_reuse(ecc_metric* m, int on);

/* One launch per small evaluation (default on).  An evaluation of at most 192 pairs -- ecc_metric_evaluate_all of up to 20
 * views, small ecc_metric_evaluate_range shards, short ecc_metric_evaluate_pairs index lists; not with use_corr, not the
 * asynchronous form -- runs as ONE kernel instead of the stream-ordered launches E1, K01, pairs, sum (the reference: two
 * kernels, two device-wide syncs and a host loop, ref: EpipolarConsistencyRadonIntermediate.cu:300-409,
 * ...RadonIntermediate.cpp:197-224; its working optimiser caller evaluates a handful of pairs per objective call, ref:
 * tools/FluoroTracking/FluoroTracking.cpp:179-211): E1 of the views whose matrix changed is computed on the host (the same
 * code, bit-identical) and travels in the kernel arguments, each workgroup fits its pair's record and samples it with four
 * waves, every value goes to the device array and -- at system scope -- to pinned host memory, the workgroup that arrives
 * last writes the word the host polls, and the host adds the values in the sum kernel's order.  Larger evaluations keep the
 * stream-ordered launches, up to 4096 pairs with several waves per pair (pairs_split_kernel), index lists without copy
 * commands (the list is read from pinned memory, the sum kernel hands the values back), and launches of at most 4096 pairs
 * also take E1 of up to 16 changed views in the arguments of their record kernel (k01_patched_kernel) instead of an e1_kernel
 * launch.  Every value and every sum has the
 * bits of the multi-launch one-wave-per-pair path (tests/test_gpu_small_eval.py, tests/test_gpu_stress_sequences.py); on = 0
 * keeps the stream-ordered launches for everything. */
int ecc_metric_set_small_eval(ecc_metric* m, int on);
/* ref: Metric::getObjectRadius (EpipolarConsistency.cpp:76-84): user value, or the estimate
 * from the FIRST projection matrix. */
int ecc_metric_get_object_radius(const ecc_metric* m, double* radius_mm);

/* All-pairs evaluate, ref: double MetricRadonIntermediate::evaluate(float* out)
 * (…RadonIntermediate.cpp:166-225) = K01 + pair kernel + host mean.  cost_nxn (host, nullable) is
 * read-modify-written exactly like the reference's `out`.  *mean = sum_pairs / n_pairs. */
/* The float64 sum over the pair values has a fixed order for a given number of values (the same bits every run), but the
 * order is a function of that number: up to 32 767 values one workgroup adds them, from 32 768 on sixteen workgroups add
 * contiguous slices and the slice sums are added in slice order.  A sum over G shards is the rank-ordered sum of G such
 * sums.  So one-device and sharded sums of the same pair values agree to float64 rounding (~1e-16), not bit for bit. */
int ecc_metric_evaluate_all(ecc_metric* m, float* cost_nxn, double* mean);
/* n_poses INDEPENDENT all-pairs evaluations of one data set (a sweep of poses, the probes of a finite-difference gradient;
 * ref: Gui/Visualization.h:59-112 plotCostFunction -- 100 steps x 6 parameters, BASELINE config 5 -- through
 * Gui/SingleImageMotion.h:84-90).  The reference evaluates them one setProjectionMatrices + evaluate at a time.
 *
 * ecc_metric_evaluate_pose_deltas: pose k = the metric's CURRENT matrices (the base: the last ecc_metric_set_projections)
 * with the views moved_views[moved_offsets[k] .. moved_offsets[k + 1]) (strictly ascending within a pose) replaced by the
 * matrices moved_Ps[12 * q ..] of the same entries q.  All poses of the call are ONE launch that lists the pairs and does E1
 * of the moved matrices, ONE record launch and ONE pair launch over the (pose, moved view) x partner grid -- n_views - 1 pairs
 * per moved view instead of n (n - 1) / 2 per pose -- and ONE segmented float64 sum that walks, per pose, the base's pair values with the pose's own
 * substituted in exactly the order the all-pairs sum adds them: every mean has the bits of ecc_metric_set_projections +
 * ecc_metric_evaluate_all on that pose's matrices (tests/test_gpu_pose_batch.py).  The base's pair values are kept between
 * calls (only the pairs of views that changed since are redone).  The metric's current matrices are unchanged by the call.
 * A pose with more than 32 moved views, or one that moves view 0 under the automatic object radius and changes it, is
 * evaluated the sequential way inside the call (same bits).  400 views of 1024^2, 600 poses of one moved view on one MI355X:
 * see DESIGN.md 4.9 (two orders of magnitude above the sequential steps: the work is 239 400 pairs, not 47 880 000).
 *
 * ecc_metric_evaluate_poses[_strided]: the same for FULL matrices per pose (Ps_batch: n_poses x n_views x 12 float64); the
 * strided form evaluates the poses first, first + stride, ... only and leaves the other entries of `means` alone (rank r of
 * N: first = r, stride = N -- poses shard with no exchange at all).  The library finds the moved views itself by comparing
 * every pose with the metric's current matrices (or with the first pose when most poses are far from those) and takes the
 * batch above; poses that are no small delta are evaluated two deep on the context's stream (pose k + 1 is handed over
 * while the device runs pose k).  Same bits either way.  The matrices of the LAST evaluated pose stay the metric's current
 * ones.  ecc_metric_set_pose_batching(m, 0): everything two deep (the launches of n_poses x (ecc_metric_set_projections +
 * ecc_metric_evaluate_all) in the same order) -- what rounds 4-5 measured.  ecc_metric_last_batched_poses: how many poses of
 * the last of these calls went through the batch. */
int ecc_metric_evaluate_pose_deltas(ecc_metric* m, int n_poses, const int32_t* moved_offsets, const int32_t* moved_views,
                                    const double* moved_Ps, double* means);
int ecc_metric_evaluate_poses(ecc_metric* m, int n_poses, const double* Ps_batch, int n_views, double* means);
int ecc_metric_evaluate_poses_strided(ecc_metric* m, int n_poses, const double* Ps_batch, int n_views, int first, int stride,
                                      double* means);
int ecc_metric_set_pose_batching(ecc_metric* m, int on);
int ecc_metric_last_batched_poses(const ecc_metric* m, int64_t* poses);

/* Multi-GPU building block: evaluate only pairs ij in [first, first+count) of the get_ij order
 * (ref: EpipolarConsistencyCommon.hxx:52-79); returns the partial sum (float64) -- the caller
 * all-reduces {sum, count}.  pair_values (host, nullable) receives `count` floats. */
int ecc_metric_evaluate_range(ecc_metric* m, int64_t first, int64_t count, float* pair_values,
                              double* partial_sum);

/* Asynchronous form for callers that own the device buffers (torch): nothing is copied to the
 * host and the call does not synchronise.  pair_values_d: `count` floats on the device
 * (nullable); sum_d: one double on the device, overwritten with the partial sum. */
int ecc_metric_evaluate_range_async(ecc_metric* m, int64_t first, int64_t count,
                                    float* pair_values_d, double* sum_d);

/* Companions of ecc_metric_evaluate_range_async for the one-process-per-GPU form: after the collective that adds the
 * ranks' partial sums has been queued on the context's stream (an RCCL all-reduce of sum_d), ecc_metric_publish_scalar
 * queues a one-thread kernel behind it that stores *value_d into the metric's pinned result slot, and
 * ecc_metric_wait_scalar polls that slot (bounded, like ecc_metric_evaluate_all's wait): the reduced value reaches the
 * host without a device-to-host copy command and its stream synchronisation. */
int ecc_metric_publish_scalar(ecc_metric* m, const double* value_d);
int ecc_metric_wait_scalar(ecc_metric* m, double* value);

/* Index-list evaluate, ref: evaluate(const std::vector<Eigen::Vector4i>&, float*)
 * (…RadonIntermediate.cpp:267-322).  idx4: n_pairs x 4 int32 on the host.  Indices are range
 * checked in every build (the reference only checks under _DEBUG, .cpp:248-264). */
int ecc_metric_evaluate_pairs(ecc_metric* m, const int32_t* idx4, int n_pairs, float* out,
                              double* mean);

/* ref: MetricRadonIntermediate::evaluateForImagePair(i, j, redundant_samples0, redundant_samples1, kappas,
 * radon_samples0, radon_samples1) (...RadonIntermediate.cpp:324-393, "visualization only"): the two
 * redundant signals of ONE pair over kappa in (-kappa_max, kappa_max) with num_samples = image diagonal
 * (:349) and the fp32-accumulated kappa loop (:367).  The reference does this on the host from read-back
 * dtrs; here one small kernel samples the device-resident dtrs.  The evident intent is implemented where
 * the reference code is unfinished: the (alpha+pi, -t) fold flips the sign of derivative
 * dtrs, sampling uses the metric's own texel rule, and *ecc = SUM (v0-v1)^2 dkappa (the reference assigns).
 * All output arrays are on the host and nullable; each holds `capacity` entries (radon_samples: 2 floats
 * per sample = texture coordinates (angle, distance) in [0,1]).  *n_samples receives the number of samples;
 * fails with ECC_ERR_INVALID_ARGUMENT (and *n_samples set) when capacity is too small --
 * ecc_metric_pair_samples_bound gives a sufficient capacity.  K01 (nullable) receives 16 floats. */
int ecc_metric_pair_samples_bound(const ecc_metric* m, int* capacity);
int ecc_metric_evaluate_for_image_pair(ecc_metric* m, int i, int j, int capacity, int* n_samples,
                                       float* redundant_samples0, float* redundant_samples1, float* kappas,
                                       float* radon_samples0, float* radon_samples1, float* K01,
                                       double* ecc);

/* Debug: the 16 K01 floats per pair the kernel used (ref: kernelEpipolarConsistencyComputeK01,
 * …RadonIntermediate.cu:13-67), for ij in [first, first+count).  Host output. */
int ecc_metric_debug_K01(ecc_metric* m, int64_t first, int64_t count, float* K01s);

/* ---- MetricDirect: consistency straight from the projection images ---------------------------- */
/* ref: class MetricDirect (LibEpipolarConsistency/EpipolarConsistencyDirect.h:28-60) and computeForImagePair
 * (EpipolarConsistencyDirect.cpp:67-219) -- no Radon intermediates; per pair 2 x n_lines line integrals through
 * the images (n_lines = twice the image diagonal unless dkappa is given), so it is ~1000x the work of
 * MetricRadonIntermediate per evaluation and meant for few views / images that change every iteration.
 * images: n x n_v x n_u float32.  on_device = 1: the metric BORROWS the device pointer (as the reference borrows
 * its textures); on_device = 0: it uploads and owns a copy.
 * use_fbcc != 0 (setFanBeamConsistency): the rectified fan-beam variant -- plain line integrals weighted by the
 * line perspectivity onto a virtual detector through the baseline (RectifiedFBCC.h, ...Direct.cpp:133-196)
 * instead of the derivative. */
typedef struct ecc_direct ecc_direct;
int ecc_direct_create(ecc_ctx* ctx, int n_images, const float* images, int on_device, int n_u, int n_v,
                      ecc_direct** out);
int ecc_direct_destroy(ecc_direct* d);
/* The metric keeps a transposed copy of the images (coalesced access for near-horizontal epipolar lines).  After
 * changing the pixels of borrowed device images call this before the next evaluation (asynchronous, a few
 * microseconds per image) -- the counterpart of re-binding the reference's textures. */
int ecc_direct_update_images(ecc_direct* d);
/* ref: MetricDirect::setProjectionMatrices; n x 12 float64 column-major. */
int ecc_direct_set_projections(ecc_direct* d, const double* Ps, int n_views);
/* ref: Metric::setObjectRadius / setEpipolarPlaneStep / MetricDirect::setFanBeamConsistency; 0 = automatic. */
int ecc_direct_set_params(ecc_direct* d, double object_radius_mm, double dkappa, int use_fbcc);
int ecc_direct_get_object_radius(const ecc_direct* d, double* radius_mm);
/* ref: double MetricDirect::evaluate(float* out) (.cpp:247-259): returns the SUM over all pairs (not the mean);
 * cost_nxn (host, nullable): entry (i,j), i<j at index i + j*n is written, the rest is preserved. */
int ecc_direct_evaluate(ecc_direct* d, float* cost_nxn, double* cost_sum);
/* ref: MetricDirect::evaluateForImagePair(i, j, redundant_samples0, redundant_samples1, kappas) (.cpp:261-272).
 * Host outputs, nullable, `capacity` entries each (lines01: 6 floats per kappa = the two epipolar lines in pixel
 * coordinates, Hessian normal form); ecc_direct_lines_bound gives a sufficient capacity.  A caller-provided
 * kappa grid (the reference accepts a non-empty `kappas` as input, .cpp:105-117) is not supported. */
int ecc_direct_lines_bound(const ecc_direct* d, int* capacity);
int ecc_direct_evaluate_for_image_pair(ecc_direct* d, int i, int j, int capacity, int* n_lines,
                                       float* redundant_samples0, float* redundant_samples1, float* kappas,
                                       float* lines01, double* metric);

/* The same with a caller-provided grid of epipolar-plane angles (the reference takes a non-empty `kappas` vector as
 * the grid, .cpp:105-117): n_kappas float32 angles on the host; outputs hold n_kappas entries each; the metric is
 * SUM (v0 - v1)^2 dkappa with the dkappa of ecc_direct_set_params (or the automatic one), as in the reference. */
int ecc_direct_evaluate_for_image_pair_kappas(ecc_direct* d, int i, int j, int n_kappas, const float* kappas_in,
                                              float* redundant_samples0, float* redundant_samples1, float* lines01,
                                              double* metric);

/* Debug: what the pair-geometry kernel fitted for pairs ij in [first, first+count): per pair
 * ECC_POLY_RECORD_FLOATS floats = { degree evaluated (4, 6, 8, 10; 0 = exact path; + 0.5 when the fit's bound says that no sample
 * of the pair can reach a clamp of the pair kernel, which then runs its clamp-free loop), x_scale, fold0, fold1 (1 = folded on the +kappa side),
 * ca[0][0..DEG+2], ca[1][...], cd[0][0..DEG+1], cd[1][...] } with DEG = 10.  Host output. */
#define ECC_POLY_RECORD_FLOATS (4 + 2 * 13 + 2 * 12)
int ecc_metric_debug_polynomials(ecc_metric* m, int64_t first, int64_t count, float* out);

/* ---- helpers shared with callers ------------------------------------------------------------ */
/* ref: get_ij (EpipolarConsistencyCommon.hxx:52-79), closed form. */
void ecc_get_ij(int64_t ij, int n, int* i, int* j);
/* Host-side E1/E5 pieces, exported for tests and for callers that shard work themselves. */
void ecc_host_pinvT(const double* P, float* PinvT12);
void ecc_host_source_position(const double* P, float* C4);
double ecc_host_object_radius(const double* P, int n_u, int n_v);

/* ref: lineToSampleDtr (EpipolarConsistencyCommon.hxx:152-171), host: line (l0, l1, l2) relative to the image
 * centre -> line[0] = angle / Pi in [0, 1], line[1] = distance in [0, 1]; returns 1 when the (alpha + Pi, -t)
 * periodicity was used (the sample's sign flips for a derivative dtr).  fp32, atan2 correctly rounded. */
int ecc_host_line_to_sample_dtr(float* line3, float range_t);

/* ref: estimateAngularRange(join_pluecker(C0, C1), radius) (EpipolarConsistency.cpp:49-59): the kappa interval
 * of epipolar planes through the baseline of P0, P1 that touch a sphere of object_radius_mm about the origin. */
void ecc_host_angular_range(const double* P0, const double* P1, double object_radius_mm, double* kappa_first,
                            double* kappa_second);
/* ref: estimateAngularStep(P0, P1, n_u, n_v) (EpipolarConsistency.cpp:61-68). */
double ecc_host_angular_step(const double* P0, const double* P1, int n_u, int n_v);
/* ref: estimateIsoCenter(Ps) (EpipolarConsistency.cpp:8-33): least-squares intersection of the principal rays;
 * O receives 4 doubles (w = 1). */
void ecc_host_iso_center(const double* Ps, int n_views, double* O);

/* ---- single-process multi-GPU: a group of devices behind the same two calls ----------------------- */
/* The reference's callers are one process, one optimiser thread: ecc->setProjectionMatrices(Ps); ecc->evaluate()
 * (ref: Gui/SingleImageMotion.h:84-90, HeaderOnly/LibOpterix/WrapNLOpt.hxx:151-173).  A group gives such a caller
 * every GPU of the node: one context, stream and host thread per device, the Radon-intermediate stack replicated on
 * every device, the pair range cut into contiguous cost-balanced shards of the get_ij order (ecc_pair_shards_balanced,
 * fixed at the first evaluation; ecc_group_metric_rebalance recomputes them), the partial sums (8 bytes per device, pinned host memory) added on the host in rank order -- the same bits on every
 * call.  Nothing is exchanged between devices on the per-evaluation path.
 * devices: n_dev HIP device indices (NULL = 0 .. n_dev-1).  An index may repeat: ranks on the same device get their
 * own stream and thread and share the caller's slabs (rehearsal of the multi-rank path on one GPU). */
typedef struct ecc_group ecc_group;
typedef struct ecc_group_metric ecc_group_metric;
int ecc_group_create(int n_dev, const int* devices, ecc_group** out);
/* Destroy group metrics first, then dtrs made from the group's contexts, then the group. */
int ecc_group_destroy(ecc_group* g);
int ecc_group_size(const ecc_group* g);
/* The context of one rank (borrowed): e.g. to produce Radon intermediates on that device yourself. */
int ecc_group_ctx(ecc_group* g, int rank, ecc_ctx** ctx);
/* Data-parallel form of ecc_radon_compute_batch for host images: rank r computes the contiguous chunk of views
 * [r*ceil(n/G), ...) on its own device, all ranks concurrently.  out[k] lives on the device that computed it;
 * ecc_group_metric_create replicates. */
int ecc_group_radon_compute_batch(ecc_group* g, const float* images, int n, int n_u, int n_v, int n_alpha, int n_t,
                                  int filter, int post_process, ecc_dtr** out);
/* rank r of `world` evaluates pairs [first, first + count) of the get_ij order: first = r*n_pairs/world (integer). */
void ecc_pair_shard(int64_t n_pairs, int world, int rank, int64_t* first, int64_t* count);
/* Cost-balanced alternative: contiguous chunks of equal MODEL COST instead of equal count -- the pair kernel's time per
 * pair grows with the pair's kappa_max, and for a circular scan the expensive pairs sit in the first rows of the pair
 * triangle (equal-count shards of an 8-rank job measured 93 ... 68 us; model for balanced shards 77.5 us each; the fit is in
 * ecc_metric_api.hip).  bounds receives world + 1 pair indices, rank r evaluates [bounds[r], bounds[r+1]).  Host, float64, a
 * function of the matrices and the object radius only (every rank of a job computes the same bounds); ~0.3 ms for 400
 * views -- once per data set.  ecc_metric_balanced_shards uses the metric's current matrices and object radius. */
int ecc_pair_shards_balanced(const double* Ps, int n_views, double object_radius_mm, int world, int64_t* bounds);
int ecc_metric_balanced_shards(ecc_metric* m, int world, int64_t* bounds);

/* ref: MetricRadonIntermediate(Ps, dtrs) (...RadonIntermediate.cpp:53-66,87-106) over a group.  dtrs may live on
 * any devices; every rank gets a replica of the whole stack (device-to-device copies, peer access where available;
 * a rank whose device already holds all of them borrows them instead) and its own ecc_metric.  Same borrowing
 * contract as ecc_metric_create. */
int ecc_group_metric_create(ecc_group* g, int n_dtrs, ecc_dtr* const* dtrs, ecc_group_metric** out);
/* Debug / test hook: groups metrics created while it is on copy the Radon-intermediate stack on every rank even when it
 * is already on the rank's device (the code path of a real multi-GPU group on a one-GPU box; doubles the memory). */
int ecc_group_debug_force_replica(int on);
int ecc_group_metric_destroy(ecc_group_metric* gm);
/* ref: setProjectionMatrices.  The matrices are copied and handed to the devices together with the next
 * evaluation (one hand-off to the rank threads per optimiser step). */
int ecc_group_metric_set_projections(ecc_group_metric* gm, const double* Ps, int n_views);
int ecc_group_metric_set_params(ecc_group_metric* gm, double object_radius_mm, double dkappa, int use_corr);
int ecc_group_metric_set_incremental(ecc_group_metric* gm, int enable);  /* every rank keeps its own shard's values */
int ecc_group_metric_set_sampling(ecc_group_metric* gm, int mode);
int ecc_group_metric_get_object_radius(ecc_group_metric* gm, double* radius_mm);
/* ref: double MetricRadonIntermediate::evaluate(float* out): all pairs, sharded over the group; *mean = sum/n_pairs.
 * cost_nxn (host, nullable): entry (i,j), i<j at index i + j*n is written, the rest preserved.  With one rank the
 * result is bit-identical to ecc_metric_evaluate_all; with G ranks it is the rank-ordered float64 sum of G shard sums. */
int ecc_group_metric_evaluate_all(ecc_group_metric* gm, float* cost_nxn, double* mean);
/* n_poses INDEPENDENT all-pairs evaluations (Ps_batch: n_poses x n_views x 12 float64; means: n_poses results): a sweep
 * of poses (ref: Gui/Visualization.h:78-98 plotCostFunction; BASELINE config 5) or the probes of a finite-difference
 * gradient.  Pose p is evaluated entirely on rank p mod G -- no exchange, G times one device's throughput; every value
 * is bit-identical to ecc_metric_evaluate_all on one device.  The matrices of the last ecc_group_metric_set_projections
 * stay the group's current ones. */
int ecc_group_metric_evaluate_poses(ecc_group_metric* gm, int n_poses, const double* Ps_batch, int n_views, double* means);
/* Recompute the cost-balanced shard boundaries from the current matrices (they are otherwise fixed at the first
 * evaluation so that repeated evaluations add the same partial sums in the same order). */
int ecc_group_metric_rebalance(ecc_group_metric* gm);
/* The per-rank metric (borrowed): few-pair calls (index lists, evaluateForImagePair) go to rank 0's metric.  Pending
 * projection matrices are handed to the devices first. */
int ecc_group_metric_rank_metric(ecc_group_metric* gm, int rank, ecc_metric** m);

/* ---- multi-process sum of the partial results (one process per GPU, one node) ------------------ */
/* The path's only exchange step is the final sum over pairs (ref: ...RadonIntermediate.cpp:216-224; SURVEY.md 8e:
 * "one all-reduce of 2 doubles per evaluation -- latency-bound").  ecc_metric_evaluate_range leaves each rank's
 * partial sum on the host; ecc_exchange_sum adds the partial sums of all ranks through a POSIX shared-memory segment
 * (a cache line per rank, polled; ~1 us) in rank order, so every rank returns the same bits.  No device is
 * involved.  All ranks call ecc_exchange_sum the same number of times.
 * name: shm name starting with '/', the same on all ranks and unique per job; rank 0 creates the segment, the other
 * ranks wait for it.  A rank that does not show up makes the others fail with ECC_ERR_UNSUPPORTED after
 * ECC_EXCHANGE_TIMEOUT_S seconds (environment, default 60) instead of hanging; after such a failure the exchange
 * stays failed (every later ecc_exchange_sum returns the error at once) -- close it, do not retry.
 * SINGLE NODE ONLY: the segment lives in this node's /dev/shm, so rank / world are the node-local rank and the number
 * of ranks on this node (LOCAL_RANK / LOCAL_WORLD_SIZE under torchrun); a job that spans nodes adds the node sums
 * with a collective of its own (sharding.open_exchange refuses WORLD_SIZE != LOCAL_WORLD_SIZE). */
/* The same exchange as an RCCL all-reduce issued by the library (one process per GPU; the exchange north_star names).  Every rank
 * owns a communicator on its context's device: rank 0 makes the 128-byte id (ecc_comm_unique_id) and the job hands it to the
 * other ranks (bench.py: torch.distributed's broadcast); ecc_comm_create is RCCL's ncclCommInitRank -- a collective, it returns
 * when all `world` ranks have called it.  ecc_metric_evaluate_range_allreduce then does one rank's share of an evaluation in
 * ONE call: pair kernel over [first, first + count) -> float64 sum on the device -> ncclAllReduce over the ranks -> the scalar
 * published to pinned host memory -> poll; all on the context's stream, no host round trip in between.  *sum_all is the sum over
 * ALL ranks' pairs (the same bits on every rank); the mean is sum_all / (n (n - 1) / 2).  All ranks call it the same number of
 * times.  RCCL is loaded at run time (dlopen of librccl.so.1: the copy the process already has -- PyTorch's -- or ROCm's);
 * without it these calls fail with ECC_ERR_UNSUPPORTED and nothing else of the library is affected.
 * ecc_comm_available: ECC_OK when RCCL can be bound in this process -- no GPU call, no collective.  A job must AGREE on it
 * across its ranks (a MIN all-reduce of the answers) before any rank calls ecc_comm_create: a rank that cannot load RCCL
 * returns from ecc_comm_create at once, and the others would wait for it inside ncclCommInitRank for ever
 * (sharding.RcclComm does this when it is given an `agree` callable; bench.py passes one). */
typedef struct ecc_comm ecc_comm;
#define ECC_COMM_ID_BYTES 128
int ecc_comm_available(void);
int ecc_comm_unique_id(void* id128);
int ecc_comm_create(ecc_ctx* ctx, const void* id128, int rank, int world, ecc_comm** out);
int ecc_comm_destroy(ecc_comm* c);
int ecc_metric_evaluate_range_allreduce(ecc_metric* m, ecc_comm* comm, int64_t first, int64_t count, double* sum_all);
#define ECC_EXCHANGE_MAX_RANKS 64
typedef struct ecc_exchange ecc_exchange;
int ecc_exchange_open(const char* name, int rank, int world, ecc_exchange** out);
int ecc_exchange_sum(ecc_exchange* ex, double partial, double* total);
int ecc_exchange_close(ecc_exchange* ex);

/* Last kernel timings measured with HIP events on the context's stream (ms), for bench.py:
 * which = 0 pair kernel of the last evaluate, 1 Radon kernel of the last radon_compute[_batch],
 * 2 pre-processing kernel of the last ecc_preprocess.
 * Timing is off by default (no events recorded); enable with ecc_ctx_enable_timing. */
int ecc_ctx_enable_timing(ecc_ctx* ctx, int enable);
int ecc_ctx_last_kernel_ms(ecc_ctx* ctx, int which, float* ms);

/* ---- experiment hooks (not part of the drop-in surface) ------------------------------------------ */
/* What scripts/ uses for A/B measurements.  Explicit calls on a handle, so that nothing outside the caller's code can
 * change the arithmetic of an evaluation.
 * ecc_debug_set_poly_tolerance: the bound (in Radon bins) on what lowering a pair's polynomial degree may cost in
 *   ECC_SAMPLING_POLYNOMIAL (DESIGN.md 4.2 "economise"); default ECC_POLY_ECONOMISE_TOL_BINS.  Changes pair values at the
 *   1e-7 level; drops the kept records / values.
 * ecc_debug_set_small_eval_bound: max_pairs >= 0 lowers the size bound of the one-launch path (192 pairs; 0 = never taken,
 *   the path's other conditions still apply); -1 restores the default.  Same bits either way.
 * ecc_debug_set_result_polling: process-wide; 0 = synchronous calls wait for the stream instead of polling the pinned
 *   result slot.  Same bits.
 * ecc_debug_set_quad_copies: ecc_ctx_set_quad_copies(ctx, on ? ECC_QUAD_COPIES_ON : ECC_QUAD_COPIES_OFF), the experiments' old name.
 * ecc_debug_small_stamps: only in builds with -DECC_SMALL_STAMPS (returns ECC_ERR_INVALID_ARGUMENT otherwise). */
#define ECC_POLY_ECONOMISE_TOL_BINS 2e-8f
int ecc_debug_set_poly_tolerance(ecc_metric* m, float tol_bins);
int ecc_debug_set_small_eval_bound(ecc_metric* m, int64_t max_pairs);
int ecc_debug_set_result_polling(int on);
int ecc_debug_set_quad_copies(ecc_ctx* ctx, int on);
/* Row-quad copies.  Metrics created from ctx AFTERWARDS keep, beside the row-paired copy of every Radon intermediate (one
 * 16-byte footprint per sample), a second copy in which four consecutive angle rows share a 128-byte line (4x the slab's
 * memory: 3.9 GB for 400 views of 768 x 768 bins).  The pairs whose baseline passes through the object (kappa_max = pi/2,
 * 3.5 % of a short scan's pairs, a fifth of its evaluation time) cross the Radon intermediates diagonally, a new angle row
 * every sample or two; the exact part of their sampling reads the row-quad copies: +2 % evaluations/s on the 400-view
 * benchmark, bit-identical values (tests/test_gpu_sampling_modes.py).  ECC_QUAD_COPIES_AUTO (default): decided ONCE per metric
 * by its first all-pairs evaluation (or shard of one) over at least 32 768 pairs, from the matrices it runs with -- built when at least 2 % of their pairs have
 * kappa_max > pi/4 (a 200-degree short scan: 3.5 %; a 90-degree scan: none, and no memory is spent) and all copies together
 * take at most a quarter of the device memory free at that time (an allocation that fails after all means "none"); that
 * evaluation is ~1.5 ms longer.  _OFF: never; _ON: always, at ecc_metric_create (offsets permitting; an allocation failure is
 * the metric's failure).
 * MEMORY a metric owns per Radon intermediate of n_alpha x n_t bins, pitch = roundup(n_t + 2, 32) floats (ecc_metric_device_bytes
 * reports the totals; the Radon intermediates themselves, (n_alpha + 2) x pitch x 4 bytes each, belong to their ecc_dtr):
 *   row-paired copy (always):  (n_alpha + 1) x pitch x 8 bytes                   768 x 768 bins: 4.92 MB, 400 views 1.97 GB
 *   row-quad copy (see above): ceil((n_alpha + 1) / 4) x pitch x 64 bytes        768 x 768 bins: 9.88 MB, 400 views 3.95 GB
 * i.e. 2x and 4x the slab (2.46 MB): a 400-view metric holds 5.9 GB beside the 0.99 GB of the stack itself -- for 0.300 -> 0.294 ms
 * per evaluation in the row-quad case.  A process that keeps many metrics alive on one device should create them from a context
 * with ECC_QUAD_COPIES_OFF (same bits, -2 %). */
#define ECC_QUAD_COPIES_AUTO (-1)
#define ECC_QUAD_COPIES_OFF 0
#define ECC_QUAD_COPIES_ON 1
int ecc_ctx_set_quad_copies(ecc_ctx* ctx, int mode);
/* Device memory the metric owns right now, in bytes (any pointer may be null): the row-paired copies, the row-quad copies
 * (0 when they were not built), everything else (geometry, per-pair records of 296 bytes, pair values, cost image, index
 * lists, the pose batch's scratch -- grown on demand, kept until the metric is destroyed). */
int ecc_metric_device_bytes(const ecc_metric* m, int64_t* paired_bytes, int64_t* quad_bytes, int64_t* other_bytes);
int ecc_debug_small_stamps(unsigned long long* out, int n_blocks);
/* Host clock (seconds, std::chrono::steady_clock) at fixed points of the metric's last ecc_metric_set_projections and last
 * synchronous all-pairs / range evaluation -- where the host's share of a step goes (scripts/step_fixed_cost.py):
 * [0] set_projections entered, [1] returned; [2] evaluate entered, [3] change detection done (first launch next),
 * [4] first pair-kernel launch returned, [5] refit / list launches queued, [6] sum launch returned, [7] result seen. */
#define ECC_STEP_STAMPS 8
int ecc_debug_step_stamps(const ecc_metric* m, double* out8);

#ifdef __cplusplus
}
#endif
#endif /* ECC_HIP_H */
