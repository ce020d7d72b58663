"""Host-side mirror of the reference's operator interface for the hot path, on top of the C ABI.

Same names, argument meaning and error behaviour as
  EpipolarConsistency::RadonIntermediate        ref: code/LibEpipolarConsistency/RadonIntermediate.h:18-128
  EpipolarConsistency::MetricRadonIntermediate  ref: code/LibEpipolarConsistency/EpipolarConsistencyRadonIntermediate.h:21-106
  EpipolarConsistency::Metric                   ref: code/LibEpipolarConsistency/EpipolarConsistency.h:49-94
(the reference is C++; the C++ adapter with the same class names is cpp/EpipolarConsistencyHip.hxx --
this module is the Python face used by tests/ and bench.py).  All arithmetic happens in
libecc_hip.so; numpy/torch only carry buffers.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (FILTER_DERIVATIVE, FILTER_NONE, FILTER_RAMP, POST_IDENTITY, POST_LOGARITHM,
                   POST_SQUARE_ROOT, EccError, check)


def pack_projection_matrices(Ps):
    """Public alias: list of 3x4 -> (n, 12) float64 column-major per view."""
    return _Ps_colmajor(Ps)


def _Ps_colmajor(Ps):
    """List/array of 3x4 matrices -> contiguous n x 12 float64, column-major per view (Eigen)."""
    A = np.asarray(Ps, dtype=np.float64).reshape(-1, 3, 4)
    return np.ascontiguousarray(A.transpose(0, 2, 1)).reshape(-1, 12)


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class Context:
    """One (device, stream) pair; replaces the reference's implicit current device + default
    stream + cudaDeviceSynchronize after each launch."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        check(_lib.lib().ecc_ctx_create(int(device), C.c_void_p(stream or 0), C.byref(self._h)))
        self.device = int(device)

    def synchronize(self):
        check(_lib.lib().ecc_ctx_synchronize(self._h))

    def enable_timing(self, on=True):
        check(_lib.lib().ecc_ctx_enable_timing(self._h, 1 if on else 0))

    def last_kernel_ms(self, which):
        """which: 'pairs', 'radon' or 'preprocess' -- HIP-event time of the last such kernel on this stream."""
        ms = C.c_float()
        check(_lib.lib().ecc_ctx_last_kernel_ms(self._h, {"pairs": 0, "radon": 1, "preprocess": 2}[which], C.byref(ms)))
        return ms.value

    def debugSetQuadCopies(self, on=True):
        """ecc_debug_set_quad_copies (experiments' old name): setQuadCopies("on" / "off")."""
        check(_lib.lib().ecc_debug_set_quad_copies(self._h, 1 if on else 0))
        return self

    def setQuadCopies(self, mode="auto"):
        """ecc_ctx_set_quad_copies: whether metrics created from this context afterwards build row-quad copies of their Radon
        intermediates (4x the slab memory; the exact part of the kappa_max = pi/2 pairs samples them; same bits): "auto"
        (default: while they fit a quarter of the free device memory), "off", "on"."""
        check(_lib.lib().ecc_ctx_set_quad_copies(self._h, {"auto": -1, "off": 0, "on": 1}[mode]))
        return self

    def setRadonArithmetic(self, mode="exact"):
        """ecc_radon_set_arithmetic: "exact" (default; unfused fp32, bit-identical to the oracle's normative variant) or
        "fma" (contracted sampling loop, bit-identical to the oracle's contracted variant, ~25 % faster)."""
        check(_lib.lib().ecc_radon_set_arithmetic(self._h, {"exact": _lib.RADON_EXACT, "fma": _lib.RADON_FMA}[mode]))
        return self

    def getRadonArithmetic(self):
        v = C.c_int()
        check(_lib.lib().ecc_radon_get_arithmetic(self._h, C.byref(v)))
        return {_lib.RADON_EXACT: "exact", _lib.RADON_FMA: "fma"}[v.value]

    def close(self):
        if self._h:
            _lib.lib().ecc_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PreProccess:  # sic, ref: struct EpipolarConsistency::PreProccess (Gui/PreProccess.h:14-50)
    """Pre-processing of X-ray projection images, same fields and defaults as the reference's struct
    (intensity.*, lowpass.*, image_geometry.*, border.*).  process() = PreProccess::process followed by
    apply_weight_cos_principal_ray when projection matrices are given (call order of
    Gui/InputDataDirect.cpp:85-86), fused into one device kernel for a whole stack."""

    class _NS:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    def __init__(self):
        self.intensity = self._NS(normalize=False, bias=0.0, scale=1.0, apply_log=False)
        self.lowpass = self._NS(gaussian_sigma=1.84, half_kernel_width=5)
        self.image_geometry = self._NS(flip_u=False, flip_v=False)
        self.border = self._NS(zero=[1, 1, 1, 1], feather=[16, 16, 16, 16], blanks=[])

    def _config(self, process=True):
        cfg = _lib.PreprocessConfig()
        _lib.lib().ecc_preprocess_defaults(C.byref(cfg))
        cfg.process = 1 if process else 0
        cfg.normalize, cfg.bias, cfg.scale = int(self.intensity.normalize), self.intensity.bias, self.intensity.scale
        cfg.apply_log = int(self.intensity.apply_log)
        cfg.gaussian_sigma, cfg.half_kernel_width = self.lowpass.gaussian_sigma, int(self.lowpass.half_kernel_width)
        cfg.flip_u, cfg.flip_v = int(self.image_geometry.flip_u), int(self.image_geometry.flip_v)
        cfg.zero = (C.c_int32 * 4)(*[int(v) for v in self.border.zero])
        cfg.feather = (C.c_int32 * 4)(*[int(v) for v in self.border.feather])
        bl = np.ascontiguousarray(np.asarray(self.border.blanks, np.int32).reshape(-1, 4))
        cfg.n_blanks = len(bl)
        cfg.blanks = bl.ctypes.data if len(bl) else None
        return cfg, bl

    def _run(self, ctx, images, Ps, out, process):
        cfg, keep = self._config(process)
        Pp = None if Ps is None else _Ps_colmajor(Ps)
        if _is_torch(images):
            import torch
            assert images.dtype == torch.float32 and images.is_contiguous() and images.is_cuda and images.dim() == 3
            out = images if out is None else out
            assert out.shape == images.shape and out.is_contiguous() and out.is_cuda
            n, n_v, n_u = images.shape
            a, b, on_dev = images.data_ptr(), out.data_ptr(), 1
        else:
            images = np.ascontiguousarray(images, np.float32)
            assert images.ndim == 3
            out = np.empty_like(images) if out is None else out
            n, n_v, n_u = images.shape
            a, b, on_dev = images.ctypes.data, out.ctypes.data, 0
        assert Pp is None or len(Pp) == n
        check(_lib.lib().ecc_preprocess(ctx._h, C.c_void_p(a), on_dev, C.c_void_p(b), n, n_u, n_v, C.byref(cfg),
                                        C.c_void_p(Pp.ctypes.data) if Pp is not None else None))
        del keep
        return out

    def process(self, ctx, images, Ps=None, out=None):
        """images: (n, n_v, n_u) float32, numpy (host; returns a new array) or torch on ctx's device (in place
        unless `out` is given).  Ps: n projection matrices -> cosine weighting is applied as well."""
        return self._run(ctx, images, Ps, out, True)

    def apply_weight_cos_principal_ray(self, ctx, images, Ps, out=None):
        """ref: PreProccess::apply_weight_cos_principal_ray alone (Gui/PreProccess.cpp:146-166)."""
        return self._run(ctx, images, Ps, out, False)


class RadonIntermediate:
    """ref: class RadonIntermediate (RadonIntermediate.h:18-128)."""
    Derivative, Ramp, None_ = FILTER_DERIVATIVE, FILTER_RAMP, FILTER_NONE
    Identity, SquareRoot, Logarithm = POST_IDENTITY, POST_SQUARE_ROOT, POST_LOGARITHM

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self._h = handle
        self._keep = None  # caller-owned device memory (torch tensor) when wrapping

    # -- constructors --------------------------------------------------------------------------
    @classmethod
    def compute(cls, ctx, image, size_alpha, size_t, filter=FILTER_DERIVATIVE, post_process=POST_IDENTITY):
        """ref: RadonIntermediate(projectionData, size_alpha, size_t, filter, post_process).
        image: (n_v, n_u) float32 numpy array (host) or torch tensor on ctx's device."""
        return cls.compute_batch(ctx, image[None] if not _is_torch(image) else image.unsqueeze(0),
                                 size_alpha, size_t, filter, post_process)[0]

    @classmethod
    def compute_batch(cls, ctx, images, size_alpha, size_t, filter=FILTER_DERIVATIVE, post_process=POST_IDENTITY):
        """images: (n, n_v, n_u) float32, numpy (host) or torch (device)."""
        if _is_torch(images):
            import torch
            assert images.dtype == torch.float32 and images.is_contiguous() and images.is_cuda
            n, n_v, n_u = images.shape
            ptr, on_dev, keep = images.data_ptr(), 1, images
        else:
            keep = np.ascontiguousarray(images, np.float32)
            n, n_v, n_u = keep.shape
            ptr, on_dev = keep.ctypes.data, 0
        hs = (C.c_void_p * n)()
        check(_lib.lib().ecc_radon_compute_batch(ctx._h, C.c_void_p(ptr), on_dev, n, n_u, n_v, size_alpha,
                                                 size_t, filter, post_process, hs))
        if on_dev:
            ctx.synchronize()  # the input tensor may be freed by the caller right after
        return [cls(ctx, C.c_void_p(h)) for h in hs]

    @classmethod
    def compute_into(cls, ctx, images, slabs, size_alpha, size_t, filter=FILTER_DERIVATIVE,
                     post_process=POST_IDENTITY):
        """Device-resident, asynchronous form: images (n, n_v, n_u) and slabs (n, slab_floats) are
        float32 torch tensors on ctx's device; returns handles that alias `slabs`."""
        n, n_v, n_u = images.shape
        assert images.is_contiguous() and slabs.is_contiguous()
        assert slabs.shape[0] == n and slabs.shape[1] == slab_floats(size_alpha, size_t)
        check(_lib.lib().ecc_radon_compute_into(ctx._h, C.c_void_p(images.data_ptr()), n, n_u, n_v, size_alpha,
                                                size_t, filter, post_process, C.c_void_p(slabs.data_ptr())))
        return [cls.wrap_device(ctx, slabs[k], size_alpha, size_t, n_u, n_v, filter) for k in range(n)]

    @classmethod
    def from_host(cls, ctx, data, n_u, n_v, filter=FILTER_DERIVATIVE):
        """ref: RadonIntermediate(const NRRD::ImageView<float>&) -- data is (n_t, n_alpha) float32."""
        data = np.ascontiguousarray(data, np.float32)
        n_t, n_alpha = data.shape
        h = C.c_void_p()
        check(_lib.lib().ecc_dtr_from_host(ctx._h, C.c_void_p(data.ctypes.data), n_alpha, n_t, n_u, n_v, filter,
                                           C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def load(cls, ctx, path):
        """ref: RadonIntermediate(const std::string path): a dtr NRRD written by the reference's tools
        (or by save()); returns (RadonIntermediate, info) where info carries the optional
        "Original Image/Projection Matrix"."""
        from . import nrrd
        data, info = nrrd.read_dtr(path)
        return cls.from_host(ctx, data, info["n_u"], info["n_v"], info["filter"]), info

    def save(self, path, projection_matrix=None):
        """ref: readback() + NRRD save with writePropertiesToMeta (RadonIntermediate.cpp:95-103,148-163)."""
        from . import nrrd
        nrrd.write_dtr(path, self.readback(), self.getOriginalImageSize(0), self.getOriginalImageSize(1),
                       self.getFilter(), projection_matrix)

    @classmethod
    def wrap_device(cls, ctx, slab, n_alpha, n_t, n_u, n_v, filter=FILTER_DERIVATIVE):
        """Adopt a torch tensor that already holds a dtr in the private layout (e.g. after an all-gather)."""
        h = C.c_void_p()
        check(_lib.lib().ecc_dtr_wrap_device(ctx._h, C.c_void_p(slab.data_ptr()), n_alpha, n_t, n_u, n_v, filter,
                                             C.byref(h)))
        r = cls(ctx, h)
        r._keep = slab
        return r

    # -- accessors -----------------------------------------------------------------------------
    def _info(self):
        a, t, u, v, f = (C.c_int() for _ in range(5))
        ba, bd = C.c_double(), C.c_double()
        check(_lib.lib().ecc_dtr_info(self._h, C.byref(a), C.byref(t), C.byref(u), C.byref(v), C.byref(f),
                                      C.byref(ba), C.byref(bd)))
        return a.value, t.value, u.value, v.value, f.value, ba.value, bd.value

    def getRadonBinNumber(self, dim):
        a, t = self._info()[:2]
        return t if dim else a

    def getOriginalImageSize(self, dim):
        u, v = self._info()[2:4]
        return v if dim else u

    def getRadonBinSize(self, dim=1):
        ba, bd = self._info()[5:7]
        return bd if dim else ba

    def getFilter(self):
        return self._info()[4]

    def isDerivative(self):
        return self.getFilter() == FILTER_DERIVATIVE

    def readback(self):
        """ref: readback() + data(): (n_t, n_alpha) float32, alpha fastest."""
        a, t = self._info()[:2]
        out = np.empty((t, a), np.float32)
        check(_lib.lib().ecc_dtr_readback(self._h, C.c_void_p(out.ctypes.data)))
        self._raw_cpu = out
        return out

    def clearRawData(self):
        """ref: clearRawData(): drop the host copy."""
        self._raw_cpu = None

    def data(self):
        """ref: data(): the host copy made by the last readback() (None before)."""
        return getattr(self, "_raw_cpu", None)

    def replaceRadonIntermediateData(self, radon_intermediate_image):
        """ref: replaceRadonIntermediateData(image) (RadonIntermediate.cpp:105-123): new dtr data from host memory,
        (n_t, n_alpha) float32, alpha fastest; original image size and filter are kept, the bin sizes follow the new
        shape.  Like in the reference a metric that already holds this object must be given the dtrs again
        (setRadonIntermediates)."""
        a = np.ascontiguousarray(radon_intermediate_image, np.float32)
        assert a.ndim == 2
        info = self._info()
        fresh = RadonIntermediate.from_host(self.ctx, a, info[2], info[3], filter=info[4])
        self.close()
        self._h, fresh._h = fresh._h, C.c_void_p()
        self._keep = None
        self._raw_cpu = a.copy()

    def tex2D(self, s, t):
        """ref: tex2D(s, t) (RadonIntermediate.h:108): HOST sample of the read-back data in texture coordinates
        [0, 1]^2 -- bilinear in binary64 on the (n - 1)-scaled grid with NRRD::ImageView's edge rule
        (HeaderOnly/NRRD/nrrd_image_view.hxx:159-205).  Call readback() first."""
        raw = self.data()
        if raw is None:
            raise ValueError("call readback() first")
        n_t, n_alpha = raw.shape
        x, y = (n_alpha - 1) * float(np.float32(s)), (n_t - 1) * float(np.float32(t))

        def cell(v, n):
            i = int(v)  # truncation, like the reference's (int)x
            f = v - i
            if i < 0:
                i, f = 0, 0.0
            if i > n - 2:
                i, f = n - 2, 1.0
            return i, f
        ix, fx = cell(x, n_alpha)
        iy, fy = cell(y, n_t)
        if fx == 0 and fy == 0:
            return float(np.float32(raw[iy, ix]))
        r = ((1.0 - fy) * ((1.0 - fx) * float(raw[iy, ix]) + fx * float(raw[iy, ix + 1]))
             + fy * ((1.0 - fx) * float(raw[iy + 1, ix]) + fx * float(raw[iy + 1, ix + 1])))
        return float(np.float32(r))

    def sample(self, line):
        """ref: sample(line) (RadonIntermediate.h:86-105): HOST sample for a line (l0, l1, l2) relative to the image
        centre; `line` (3 floats) is overwritten with the sample location like in the reference.  Returns the value,
        negated on the folded branch of a derivative dtr -- the evident intent, as for evaluateForImagePair: the
        reference's own flip test comes after lineToSampleDtr has already folded the angle and can never fire
        (RadonIntermediate.h:91-100)."""
        l = np.ascontiguousarray(line, np.float32).reshape(3).copy()
        range_t = np.float32(self.getRadonBinSize(1)) * np.float32(self.getRadonBinNumber(1))
        folded = _lib.lib().ecc_host_line_to_sample_dtr(C.c_void_p(l.ctypes.data), C.c_float(float(range_t)))
        try:
            line[:3] = l
        except TypeError:
            pass
        v = self.tex2D(l[0], l[1])
        return -v if (folded and self.isDerivative()) else v

    def device_view(self):
        base, pitch, rows = C.c_void_p(), C.c_int(), C.c_int()
        check(_lib.lib().ecc_dtr_device_view(self._h, C.byref(base), C.byref(pitch), C.byref(rows)))
        return base.value, pitch.value, rows.value

    def close(self):
        if self._h and self.ctx._h:  # after the context is gone its objects must not be touched any more
            _lib.lib().ecc_dtr_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MetricRadonIntermediate:
    """ref: class MetricRadonIntermediate : public Metric."""

    def __init__(self, ctx, Ps=None, dtrs=None):
        self.ctx = ctx
        self._h = C.c_void_p()
        self._dtrs = []
        self._Ps = None
        self._params = [0.0, 0.0, 0]
        d = type(self).default_sampling
        self._sampling = 0 if d is None else (self._SAMPLING[d] if isinstance(d, str) else int(d))
        self._incremental = False
        self._record_reuse = None  # library default (on unless ECC_RECORD_REUSE=0)
        self._small_eval = None    # library default (on)
        self._debug = {}           # ecc_debug_* settings of this object (experiments), re-applied to a new handle
        # the optimiser loop's two calls, setProjectionMatrices + evaluate(): data pointers of the caller's (n, 12) arrays by
        # array object (ndarray.ctypes costs 1.2 us per call; the arrays are kept alive here, at most 1024 of them), and the
        # result cell of evaluate()
        self._ptr_cache = {}
        self._mean = C.c_double()
        self._mean_ref = C.byref(self._mean)
        if dtrs is not None:
            self.setRadonIntermediates(dtrs)
        if Ps is not None:
            self.setProjectionMatrices(Ps)

    def setRadonIntermediates(self, dtrs):
        if self._h:
            _lib.lib().ecc_metric_destroy(self._h)
            self._h = C.c_void_p()
        self._dtrs = list(dtrs)  # borrowed, kept alive here
        hs = (C.c_void_p * len(self._dtrs))(*[d._h for d in self._dtrs])
        check(_lib.lib().ecc_metric_create(self.ctx._h, len(self._dtrs), hs, C.byref(self._h)))
        check(_lib.lib().ecc_metric_set_params(self._h, *self._params))
        check(_lib.lib().ecc_metric_set_sampling(self._h, self._sampling))
        check(_lib.lib().ecc_metric_set_incremental(self._h, int(self._incremental)))
        if self._record_reuse is not None:
            check(_lib.lib().ecc_metric_set_record_reuse(self._h, int(self._record_reuse)))
        if self._small_eval is not None:
            check(_lib.lib().ecc_metric_set_small_eval(self._h, int(self._small_eval)))
        self._apply_debug()
        if self._Ps is not None:
            self.setProjectionMatrices(self._Ps)
        return self

    def _apply_debug(self):
        if "poly_tolerance" in self._debug:
            check(_lib.lib().ecc_debug_set_poly_tolerance(self._h, float(self._debug["poly_tolerance"])))
        if "small_eval_bound" in self._debug:
            check(_lib.lib().ecc_debug_set_small_eval_bound(self._h, int(self._debug["small_eval_bound"])))

    def debugSetPolyTolerance(self, tol_bins):
        """ecc_debug_set_poly_tolerance (experiments): economisation bound of the polynomial path in Radon bins."""
        self._debug["poly_tolerance"] = float(tol_bins)
        if self._h:
            self._apply_debug()
        return self

    def debugSetSmallEvalBound(self, max_pairs):
        """ecc_debug_set_small_eval_bound (experiments): size bound of the one-launch path, -1 = default."""
        self._debug["small_eval_bound"] = int(max_pairs)
        if self._h:
            self._apply_debug()
        return self

    def getRadonIntermediates(self):
        return self._dtrs

    def refreshRadonIntermediates(self, first=0, count=None):
        """Not in the reference (ecc_metric_refresh_dtrs): the metric samples private row-paired copies of its dtrs;
        after the slabs behind dtrs [first, first+count) were recomputed in place, re-copy them (asynchronous)."""
        count = len(self._dtrs) - first if count is None else count
        check(_lib.lib().ecc_metric_refresh_dtrs(self._h, int(first), int(count)))
        return self

    def setProjectionMatrices(self, Ps):
        """Ps: list of 3x4 matrices, or (fast path) an (n, 12) float64 C-contiguous array that is
        already column-major per view (what Eigen's Ps[i].data() holds)."""
        hit = self._ptr_cache.get(id(Ps))
        if hit is not None and hit[0] is Ps and hit[2] == Ps.shape:  # this very array object has passed the checks below
            self._Ps = Ps
            if self._h:
                check(_lib.lib().ecc_metric_set_projections(self._h, hit[1], hit[2][0]))
            return self
        if isinstance(Ps, np.ndarray) and Ps.ndim == 2 and Ps.shape[1] == 12 and Ps.dtype == np.float64 \
                and Ps.flags["C_CONTIGUOUS"]:
            self._Ps = Ps
            if len(self._ptr_cache) >= 1024:
                self._ptr_cache.clear()
            self._ptr_cache[id(Ps)] = (Ps, C.c_void_p(Ps.ctypes.data), Ps.shape)  # (an ndarray's buffer does not move while it is referenced)
        else:
            self._Ps = _Ps_colmajor(Ps)
        if self._h:
            check(_lib.lib().ecc_metric_set_projections(self._h, C.c_void_p(self._Ps.ctypes.data), len(self._Ps)))
        return self

    def getProjectionMatrices(self):
        return [] if self._Ps is None else [p.reshape(4, 3).T.copy() for p in self._Ps]

    def getNumberOfProjetions(self):  # sic, ref: ...RadonIntermediate.h:64
        return 0 if self._Ps is None else len(self._Ps)

    def _push_params(self):
        if self._h:
            check(_lib.lib().ecc_metric_set_params(self._h, *self._params))

    def setObjectRadius(self, radius_mm=0.0):
        self._params[0] = float(radius_mm)
        self._push_params()
        return self

    def getObjectRadius(self):
        r = C.c_double()
        check(_lib.lib().ecc_metric_get_object_radius(self._h, C.byref(r)))
        return r.value

    def setEpipolarPlaneStep(self, dkappa_rad=0.0):
        self._params[1] = float(dkappa_rad)
        self._push_params()
        return self

    setdKappa = setEpipolarPlaneStep

    def useCorrelation(self, corr=True):
        self._params[2] = 1 if corr else 0
        self._push_params()
        return self

    # class-wide default for new objects: None = the library's default (ECC_SAMPLING_AUTO)
    default_sampling = None
    _SAMPLING = {"auto": 0, "polynomial": 1, "per_sample": 2, "reference": 3}

    def setSampling(self, mode="auto"):
        """Not in the reference (ecc_metric_set_sampling): "auto" (reference arithmetic for evaluations of at most
        512 pairs, fitted polynomials above), "polynomial", "per_sample" or "reference"."""
        self._sampling = self._SAMPLING[mode] if isinstance(mode, str) else int(mode)
        if self._h:
            check(_lib.lib().ecc_metric_set_sampling(self._h, self._sampling))
        return self

    def setIncremental(self, enable=True):
        """Not in the reference (ecc_metric_set_incremental): keep the pair values of the last all-pairs / range
        evaluation on the device and re-evaluate only the pairs of views whose matrix changed since -- the optimiser
        pattern of Gui/SingleImageMotion.h (one view moves per call).  Results are bit-identical to full evaluations."""
        self._incremental = bool(enable)
        if self._h:
            check(_lib.lib().ecc_metric_set_incremental(self._h, int(self._incremental)))
        return self

    def setRecordReuse(self, on=True, always=False):
        """Not in the reference (ecc_metric_set_record_reuse, default on): keep the per-pair geometry records of the last
        all-pairs / range evaluation and refit only the pairs whose matrices changed; every pair is still sampled and
        results are bit-identical either way.  on=True: for ranges of more than 4096 pairs (smaller evaluations are
        faster refitting everything); always=True: for every size."""
        self._record_reuse = (2 if always else 1) if on else 0
        if self._h:
            check(_lib.lib().ecc_metric_set_record_reuse(self._h, int(self._record_reuse)))
        return self

    def setSmallEval(self, on=True):
        """ecc_metric_set_small_eval: evaluations of at most 192 pairs (ECC_SMALL_EVAL_MAX_PAIRS) as ONE launch, and E1 in
        the record kernel's arguments for launches of at most 4096 pairs (default on; bit-identical results).  Kept
        across setRadonIntermediates like the other settings."""
        self._small_eval = 1 if on else 0
        if self._h:
            check(_lib.lib().ecc_metric_set_small_eval(self._h, self._small_eval))
        return self

    def device_bytes(self):
        """ecc_metric_device_bytes -> {"paired_copies", "quad_copies", "other"}: device memory this metric owns right now (the
        Radon intermediates themselves belong to their RadonIntermediate objects)."""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(_lib.lib().ecc_metric_device_bytes(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"paired_copies": a.value, "quad_copies": b.value, "other": c.value}

    def last_evaluated_pairs(self):
        """Pairs the last evaluate() / evaluate_range() actually recomputed."""
        v = C.c_int64()
        check(_lib.lib().ecc_metric_last_evaluated_pairs(self._h, C.byref(v)))
        return v.value

    # -- evaluation ----------------------------------------------------------------------------
    def evaluate(self, arg=None, out=None):
        """evaluate()                      -> mean over all pairs
        evaluate(cost)  cost: (n,n) float32 C-contiguous array, entry [j, i] (= index i + j*n), i<j, is written
        evaluate(set_of_views[, out])   -> mean over all pairs inside the subset
        evaluate(indices[, out])        -> mean over explicit (P0,P1,dtr0,dtr1) tuples
        ref: ...RadonIntermediate.cpp:166-225, :228-245, :267-322."""
        L = _lib.lib()
        if arg is None:
            check(L.ecc_metric_evaluate_all(self._h, None, self._mean_ref))
            return self._mean.value
        mean = C.c_double()
        if arg is None or (isinstance(arg, np.ndarray) and arg.dtype == np.float32 and arg.ndim == 2
                           and out is None and arg.shape[0] == arg.shape[1] == self.getNumberOfProjetions()):
            cost = arg
            if cost is not None and not cost.flags["C_CONTIGUOUS"]:
                raise ValueError("cost image must be C-contiguous float32")
            check(L.ecc_metric_evaluate_all(self._h, C.c_void_p(cost.ctypes.data if cost is not None else 0),
                                            C.byref(mean)))
            return mean.value
        if isinstance(arg, (set, frozenset)):
            views = sorted(arg)
            idx = [(a, b, a, b) for k, a in enumerate(views) for b in views[k + 1:]]
        else:
            idx = arg
        hit_i, hit_o = self._ptr_cache.get(id(idx)), self._ptr_cache.get(id(out))
        if hit_i is not None and hit_o is not None and hit_i[0] is idx and hit_o[0] is out and hit_i[2] == idx.shape \
                and hit_o[2] == out.shape:  # these very arrays have passed the checks below
            check(L.ecc_metric_evaluate_pairs(self._h, hit_i[1], hit_i[2][0], hit_o[1], self._mean_ref))
            return self._mean.value
        cacheable = isinstance(idx, np.ndarray) and idx.dtype == np.int32 and idx.ndim == 2 and idx.shape[1] == 4 \
            and idx.flags["C_CONTIGUOUS"] and isinstance(out, np.ndarray) and out.flags["C_CONTIGUOUS"]
        idx0 = idx
        idx = np.ascontiguousarray(idx, np.int32).reshape(-1, 4)
        if out is None:
            out = np.empty(len(idx), np.float32)
        assert out.dtype == np.float32 and out.size >= len(idx)
        pi, po = C.c_void_p(idx.ctypes.data), C.c_void_p(out.ctypes.data)
        if cacheable:  # (idx is a view of idx0 then: the same buffer)
            if len(self._ptr_cache) >= 1024:
                self._ptr_cache.clear()
            self._ptr_cache[id(idx0)] = (idx0, pi, idx0.shape)
            self._ptr_cache[id(out)] = (out, po, out.shape)
        check(L.ecc_metric_evaluate_pairs(self._h, pi, len(idx), po, C.byref(mean)))
        return mean.value

    def evaluate_poses(self, poses, first=0, stride=1):
        """ecc_metric_evaluate_poses[_strided]: independent all-pairs evaluations of several poses on this metric; poses: a
        sequence of (n, 12) column-major arrays (pack_projection_matrices) or lists of 3x4 matrices; returns the means
        (bit-identical to evaluating them one by one).  Poses that differ from the current matrices (or from the first pose)
        in a few views go through ONE batched record / pair / sum launch each (setPoseBatching(False): all of them two deep on
        the stream, as in rounds 4-5).  first / stride: only the poses first, first + stride, ... (the others' means stay 0).
        The last evaluated pose's matrices stay current."""
        if isinstance(poses, np.ndarray) and poses.ndim == 3 and poses.shape[2] == 12 and poses.dtype == np.float64 \
                and poses.flags["C_CONTIGUOUS"]:
            flat = poses  # (K, n, 12) already packed: no copy (600 poses of 400 views are 23 MB)
        else:
            flat = np.ascontiguousarray(np.stack([p if (isinstance(p, np.ndarray) and p.ndim == 2 and p.shape[1] == 12)
                                                  else _Ps_colmajor(p) for p in poses]), np.float64)
        means = np.zeros(len(flat), np.float64)
        check(_lib.lib().ecc_metric_evaluate_poses_strided(self._h, len(flat), C.c_void_p(flat.ctypes.data), flat.shape[1],
                                                           int(first), int(stride), C.c_void_p(means.ctypes.data)))
        mine = range(int(first), len(flat), int(stride))
        if len(mine):
            self._Ps = flat[mine[-1]]
        return means

    def evaluate_pose_deltas(self, moved_views, moved_Ps):
        """ecc_metric_evaluate_pose_deltas: pose k = the current matrices with the views moved_views[k] (a sequence of view
        indices, strictly ascending) replaced by moved_Ps[k] (the same number of 3x4 matrices, or (c, 12) column-major rows).
        Returns the means, every one bit-identical to setProjectionMatrices + evaluate of that pose; the current matrices stay."""
        off = [0]
        views = []
        rows = []
        for vk, Pk in zip(moved_views, moved_Ps):
            vk = [int(v) for v in np.atleast_1d(vk)]
            Pk = np.asarray(Pk, np.float64)
            Pk = Pk.reshape(len(vk), 12) if (Pk.shape[-1] == 12 and Pk.ndim <= 2) else _Ps_colmajor(Pk)
            views += vk
            rows.append(Pk)
            off.append(len(views))
        flat = np.concatenate(rows) if rows and len(views) else np.zeros((0, 12))
        return self.evaluate_pose_deltas_packed(off, views, flat)

    def evaluate_pose_deltas_packed(self, moved_offsets, moved_views, moved_Ps):
        """The C signature itself: moved_offsets (K + 1 int32, [0] = 0), moved_views (Q int32, ascending within a pose), moved_Ps
        ((Q, 12) float64, column-major per matrix).  No per-pose Python work."""
        off = np.ascontiguousarray(moved_offsets, np.int32)
        views = np.ascontiguousarray(moved_views, np.int32)
        flat = np.ascontiguousarray(moved_Ps, np.float64).reshape(-1, 12)
        if len(off) < 1 or int(off[-1]) != len(views) or len(flat) != len(views):
            raise ValueError("moved_offsets / moved_views / moved_Ps disagree")
        means = np.zeros(len(off) - 1, np.float64)
        check(_lib.lib().ecc_metric_evaluate_pose_deltas(self._h, len(means), C.c_void_p(off.ctypes.data),
                                                         C.c_void_p(views.ctypes.data) if len(views) else None,
                                                         C.c_void_p(flat.ctypes.data) if len(views) else None,
                                                         C.c_void_p(means.ctypes.data)))
        return means

    def setPoseBatching(self, on=True):
        """ecc_metric_set_pose_batching: off = evaluate_poses runs every pose as its own stream-ordered evaluation."""
        check(_lib.lib().ecc_metric_set_pose_batching(self._h, 1 if on else 0))
        return self

    def last_batched_poses(self):
        v = C.c_int64(0)
        check(_lib.lib().ecc_metric_last_batched_poses(self._h, C.byref(v)))
        return v.value

    def evaluateForImagePair(self, i, j):
        """ref: evaluateForImagePair(i, j, redundant_samples0, redundant_samples1, kappas, radon_samples0,
        radon_samples1) (...RadonIntermediate.cpp:324-393, visualisation).  Returns (ecc, dict) with the
        reference's five output vectors (radon_samples as (n, 2) arrays of (angle, distance) texture
        coordinates) and K01."""
        L = _lib.lib()
        cap = C.c_int()
        check(L.ecc_metric_pair_samples_bound(self._h, C.byref(cap)))
        cap = cap.value
        s0, s1, kap = (np.empty(cap, np.float32) for _ in range(3))
        r0, r1 = np.empty((cap, 2), np.float32), np.empty((cap, 2), np.float32)
        K01 = np.empty(16, np.float32)
        n, ecc = C.c_int(), C.c_double()
        check(L.ecc_metric_evaluate_for_image_pair(
            self._h, int(i), int(j), cap, C.byref(n), C.c_void_p(s0.ctypes.data), C.c_void_p(s1.ctypes.data),
            C.c_void_p(kap.ctypes.data), C.c_void_p(r0.ctypes.data), C.c_void_p(r1.ctypes.data),
            C.c_void_p(K01.ctypes.data), C.byref(ecc)))
        n = n.value
        return ecc.value, dict(redundant_samples0=s0[:n], redundant_samples1=s1[:n], kappas=kap[:n],
                               radon_samples0=r0[:n], radon_samples1=r1[:n], K01=K01)

    def balanced_shards(self, world):
        """ecc_metric_balanced_shards: cost-balanced shard boundaries for the current matrices and object radius."""
        b = np.zeros(int(world) + 1, np.int64)
        check(_lib.lib().ecc_metric_balanced_shards(self._h, int(world), C.c_void_p(b.ctypes.data)))
        return [int(v) for v in b]

    def evaluate_range(self, first, count, want_pairs=False):
        """Partial sum over pairs [first, first+count) of the get_ij order (multi-GPU shard)."""
        s = C.c_double()
        vals = np.empty(count, np.float32) if want_pairs else None
        check(_lib.lib().ecc_metric_evaluate_range(self._h, int(first), int(count),
                                                   C.c_void_p(vals.ctypes.data if want_pairs else 0), C.byref(s)))
        return (s.value, vals) if want_pairs else s.value

    def evaluate_range_allreduce(self, comm, first, count):
        """ecc_metric_evaluate_range_allreduce: this rank's pairs, then the RCCL all-reduce of the partial sums over the ranks of
        `comm` (sharding.RcclComm), all stream-ordered inside one call; returns the sum over ALL ranks' pairs."""
        check(_lib.lib().ecc_metric_evaluate_range_allreduce(self._h, comm._h, int(first), int(count), self._mean_ref))
        return self._mean.value

    def evaluate_range_async(self, first, count, sum_tensor, pair_tensor=None):
        """Device-resident, non-synchronising form: sum_tensor is a 1-element float64 torch tensor."""
        check(_lib.lib().ecc_metric_evaluate_range_async(
            self._h, int(first), int(count), C.c_void_p(pair_tensor.data_ptr() if pair_tensor is not None else 0),
            C.c_void_p(sum_tensor.data_ptr())))

    def publish_scalar(self, value_tensor):
        """ecc_metric_publish_scalar: queue the hand-over of a 1-element float64 device tensor (e.g. an all-reduced partial
        sum) to the metric's pinned result slot on the context's stream; wait_scalar() returns it."""
        check(_lib.lib().ecc_metric_publish_scalar(self._h, C.c_void_p(value_tensor.data_ptr())))

    def wait_scalar(self):
        v = C.c_double()
        check(_lib.lib().ecc_metric_wait_scalar(self._h, C.byref(v)))
        return v.value

    def debug_geometry(self):
        n = self.getNumberOfProjetions()
        PinvTs, Cs = np.empty((n, 12), np.float32), np.empty((n, 4), np.float32)
        check(_lib.lib().ecc_metric_debug_geometry(self._h, C.c_void_p(PinvTs.ctypes.data), C.c_void_p(Cs.ctypes.data)))
        return PinvTs, Cs

    def debug_K01(self, first, count):
        out = np.empty((count, 16), np.float32)
        check(_lib.lib().ecc_metric_debug_K01(self._h, int(first), int(count), C.c_void_p(out.ctypes.data)))
        return out

    def debug_polynomials(self, first, count):
        """The sample-coordinate polynomials the pair-geometry kernel fitted: list of dicts with
        poly_ok, degree (the pair kernel evaluates the polynomials up to it), clamp_free (no sample can reach a clamp), x_scale, fold (2,), ca (2, 13), cd (2, 12)."""
        n = 4 + 2 * 13 + 2 * 12
        out = np.empty((count, n), np.float32)
        check(_lib.lib().ecc_metric_debug_polynomials(self._h, int(first), int(count), C.c_void_p(out.ctypes.data)))
        return [dict(poly_ok=bool(r[0]), degree=int(r[0]), clamp_free=bool(r[0] != int(r[0])), x_scale=float(r[1]), fold=r[2:4] > 0, ca=r[4:30].reshape(2, 13).astype(np.float64),
                     cd=r[30:54].reshape(2, 12).astype(np.float64)) for r in out]

    def close(self):
        if self._h and self.ctx._h:
            _lib.lib().ecc_metric_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _BorrowedContext:
    """A context owned by a Group (never destroyed from here)."""

    def __init__(self, handle, device, group):
        self._h, self.device, self._group = handle, device, group  # keeps the group (and so the context) alive

    def synchronize(self):
        check(_lib.lib().ecc_ctx_synchronize(self._h))

    def close(self):
        self._h = C.c_void_p()


class Group:
    """Single-process multi-GPU: one context, stream and host thread per device (ecc_group_* of the C ABI).
    devices: list of HIP device indices (an index may repeat: rehearsal of the multi-rank path on one GPU)."""

    def __init__(self, devices):
        devices = [int(d) for d in devices]
        arr = (C.c_int * len(devices))(*devices)
        self._h = C.c_void_p()
        check(_lib.lib().ecc_group_create(len(devices), arr, C.byref(self._h)))
        self.devices = devices
        self._ctxs = {}

    def __len__(self):
        return _lib.lib().ecc_group_size(self._h)

    def context(self, rank):
        if rank not in self._ctxs:
            h = C.c_void_p()
            check(_lib.lib().ecc_group_ctx(self._h, int(rank), C.byref(h)))
            self._ctxs[rank] = _BorrowedContext(h, self.devices[rank], self)
        return self._ctxs[rank]

    def compute_batch(self, images, size_alpha, size_t, filter=FILTER_DERIVATIVE, post_process=POST_IDENTITY):
        """Radon intermediates of host images (n, n_v, n_u), data-parallel over the group's devices."""
        keep = np.ascontiguousarray(images, np.float32)
        n, n_v, n_u = keep.shape
        hs = (C.c_void_p * n)()
        check(_lib.lib().ecc_group_radon_compute_batch(self._h, C.c_void_p(keep.ctypes.data), n, n_u, n_v, size_alpha,
                                                       size_t, filter, post_process, hs))
        chunk = (n + len(self.devices) - 1) // len(self.devices)
        ctxs = [self.context(r) for r in range(len(self.devices))]
        return [RadonIntermediate(ctxs[k // chunk], C.c_void_p(h)) for k, h in enumerate(hs)]

    def close(self):
        """Destroy group metrics and Radon intermediates made from the group's contexts first."""
        if self._h:
            for c in self._ctxs.values():
                c.close()  # objects that still hold one of these contexts will not touch it any more
            self._ctxs = {}
            _lib.lib().ecc_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pair_shard(n_pairs, world, rank):
    """ecc_pair_shard: (first, count) of `rank`'s contiguous chunk of the get_ij order."""
    a, b = C.c_int64(), C.c_int64()
    _lib.lib().ecc_pair_shard(int(n_pairs), int(world), int(rank), C.byref(a), C.byref(b))
    return a.value, b.value


def pair_shards_balanced(Ps, object_radius_mm, world):
    """ecc_pair_shards_balanced: world + 1 boundaries of contiguous, cost-balanced chunks of the get_ij order (rank r
    evaluates [b[r], b[r+1])).  Ps: list of 3x4 matrices or an (n, 12) column-major array."""
    if isinstance(Ps, np.ndarray) and Ps.ndim == 2 and Ps.shape[1] == 12:
        flat = np.ascontiguousarray(Ps, np.float64)
    else:
        flat = _Ps_colmajor(Ps)
    b = np.zeros(int(world) + 1, np.int64)
    check(_lib.lib().ecc_pair_shards_balanced(C.c_void_p(flat.ctypes.data), len(flat), float(object_radius_mm), int(world),
                                              C.c_void_p(b.ctypes.data)))
    return [int(v) for v in b]


class GroupMetricRadonIntermediate:
    """MetricRadonIntermediate over a Group: the same setProjectionMatrices / evaluate calls, every device of the
    group evaluating a contiguous shard of the pair range (ecc_group_metric_* of the C ABI)."""

    def __init__(self, group, Ps=None, dtrs=None):
        self.group = group
        self._h = C.c_void_p()
        self._dtrs = []
        self._Ps = None
        if dtrs is not None:
            self.setRadonIntermediates(dtrs)
        if Ps is not None:
            self.setProjectionMatrices(Ps)

    def setRadonIntermediates(self, dtrs):
        self.close()
        self._dtrs = list(dtrs)
        hs = (C.c_void_p * len(self._dtrs))(*[d._h for d in self._dtrs])
        check(_lib.lib().ecc_group_metric_create(self.group._h, len(self._dtrs), hs, C.byref(self._h)))
        d = MetricRadonIntermediate.default_sampling
        if d is not None:
            self.setSampling(d)
        if self._Ps is not None:
            self.setProjectionMatrices(self._Ps)
        return self

    def setProjectionMatrices(self, Ps):
        if isinstance(Ps, np.ndarray) and Ps.ndim == 2 and Ps.shape[1] == 12 and Ps.dtype == np.float64 \
                and Ps.flags["C_CONTIGUOUS"]:
            self._Ps = Ps
        else:
            self._Ps = _Ps_colmajor(Ps)
        if self._h:
            check(_lib.lib().ecc_group_metric_set_projections(self._h, C.c_void_p(self._Ps.ctypes.data), len(self._Ps)))
        return self

    def getNumberOfProjetions(self):  # sic
        return 0 if self._Ps is None else len(self._Ps)

    def setObjectRadius(self, radius_mm=0.0, dkappa=0.0, use_corr=False):
        check(_lib.lib().ecc_group_metric_set_params(self._h, float(radius_mm), float(dkappa), 1 if use_corr else 0))
        return self

    def getObjectRadius(self):
        r = C.c_double()
        check(_lib.lib().ecc_group_metric_get_object_radius(self._h, C.byref(r)))
        return r.value

    def setSampling(self, mode="auto"):
        mode = MetricRadonIntermediate._SAMPLING[mode] if isinstance(mode, str) else int(mode)
        check(_lib.lib().ecc_group_metric_set_sampling(self._h, mode))
        return self

    def setIncremental(self, enable=True):
        """ecc_group_metric_set_incremental: every rank keeps the pair values of its shard (see
        MetricRadonIntermediate.setIncremental)."""
        check(_lib.lib().ecc_group_metric_set_incremental(self._h, int(bool(enable))))
        return self

    def last_evaluated_pairs(self):
        """Pairs the ranks recomputed in the last evaluation, summed over the ranks."""
        total = 0
        for r in range(len(self.group)):
            m, v = C.c_void_p(), C.c_int64()
            check(_lib.lib().ecc_group_metric_rank_metric(self._h, r, C.byref(m)))
            check(_lib.lib().ecc_metric_last_evaluated_pairs(m, C.byref(v)))
            total += v.value
        return total

    def evaluate_poses(self, poses):
        """Independent all-pairs evaluations of several poses (ecc_group_metric_evaluate_poses): poses is a sequence of
        (n, 12) column-major arrays (pack_projection_matrices) or lists of 3x4 matrices; returns the means.  Pose p runs
        entirely on rank p mod G."""
        flat = np.ascontiguousarray(np.stack([p if (isinstance(p, np.ndarray) and p.ndim == 2 and p.shape[1] == 12)
                                              else _Ps_colmajor(p) for p in poses]), np.float64)
        means = np.zeros(len(flat), np.float64)
        check(_lib.lib().ecc_group_metric_evaluate_poses(self._h, len(flat), C.c_void_p(flat.ctypes.data), flat.shape[1],
                                                         C.c_void_p(means.ctypes.data)))
        return means

    def rebalance(self):
        """Recompute the cost-balanced shard boundaries at the next evaluation (ecc_group_metric_rebalance)."""
        check(_lib.lib().ecc_group_metric_rebalance(self._h))
        return self

    def evaluate(self, cost=None):
        mean = C.c_double()
        if cost is not None:
            n = self.getNumberOfProjetions()
            assert cost.dtype == np.float32 and cost.flags["C_CONTIGUOUS"] and cost.shape == (n, n)
        check(_lib.lib().ecc_group_metric_evaluate_all(self._h, C.c_void_p(cost.ctypes.data if cost is not None else 0),
                                                       C.byref(mean)))
        return mean.value

    def close(self):
        if self._h and self.group._h:
            _lib.lib().ecc_group_metric_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MetricDirect:
    """ref: class MetricDirect : public Metric (EpipolarConsistencyDirect.h:28-60): epipolar consistency straight
    from the projection images.  images: (n, n_v, n_u) float32, numpy (uploaded, owned) or a torch tensor on ctx's
    device (borrowed, like the reference borrows its textures)."""

    def __init__(self, ctx, Ps, images):
        self.ctx = ctx
        self._h = C.c_void_p()
        self._params = [0.0, 0.0, 0]
        self._Ps = None
        self.setProjectionImages(images)
        if Ps is not None:
            self.setProjectionMatrices(Ps)

    def setProjectionImages(self, images):
        if self._h:
            _lib.lib().ecc_direct_destroy(self._h)
            self._h = C.c_void_p()
        if _is_torch(images):
            import torch
            assert images.dtype == torch.float32 and images.is_contiguous() and images.is_cuda and images.dim() == 3
            self._keep, ptr, on_dev = images, images.data_ptr(), 1
        else:
            self._keep = np.ascontiguousarray(images, np.float32)
            assert self._keep.ndim == 3
            ptr, on_dev = self._keep.ctypes.data, 0
        n, n_v, n_u = self._keep.shape
        self._n = n
        check(_lib.lib().ecc_direct_create(self.ctx._h, n, C.c_void_p(ptr), on_dev, n_u, n_v, C.byref(self._h)))
        check(_lib.lib().ecc_direct_set_params(self._h, *self._params))
        if self._Ps is not None:
            self.setProjectionMatrices(self._Ps)
        return self

    def setProjectionMatrices(self, Ps):
        self._Ps = _Ps_colmajor(Ps)
        check(_lib.lib().ecc_direct_set_projections(self._h, C.c_void_p(self._Ps.ctypes.data), len(self._Ps)))
        return self

    def getNumberOfProjetions(self):  # sic
        return self._n

    def updateImages(self):
        """Call after changing the pixels of borrowed device images (refreshes the transposed copy)."""
        check(_lib.lib().ecc_direct_update_images(self._h))
        return self

    def _push(self):
        check(_lib.lib().ecc_direct_set_params(self._h, *self._params))
        return self

    def setObjectRadius(self, radius_mm=0.0):
        self._params[0] = float(radius_mm)
        return self._push()

    def getObjectRadius(self):
        r = C.c_double()
        check(_lib.lib().ecc_direct_get_object_radius(self._h, C.byref(r)))
        return r.value

    def setEpipolarPlaneStep(self, dkappa_rad=0.0):
        self._params[1] = float(dkappa_rad)
        return self._push()

    def setFanBeamConsistency(self, fbcc=True):
        self._params[2] = 1 if fbcc else 0
        return self._push()

    def evaluate(self, cost=None):
        """ref: MetricDirect::evaluate(float* out): the SUM over all pairs; cost (n,n) float32: entry [j, i], i<j."""
        s = C.c_double()
        if cost is not None:
            assert cost.dtype == np.float32 and cost.flags["C_CONTIGUOUS"] and cost.shape == (self._n, self._n)
        check(_lib.lib().ecc_direct_evaluate(self._h, C.c_void_p(cost.ctypes.data if cost is not None else 0), C.byref(s)))
        return s.value

    def evaluateForImagePair(self, i, j, kappas=None):
        """ref: evaluateForImagePair(i, j, redundant_samples0, redundant_samples1, kappas); also returns the lines.
        kappas: optional caller-provided grid of plane angles (the reference takes a non-empty `kappas` as input)."""
        L = _lib.lib()
        if kappas is not None:
            kap = np.ascontiguousarray(kappas, np.float32).reshape(-1)
            s0, s1, lines = np.empty(len(kap), np.float32), np.empty(len(kap), np.float32), np.empty((len(kap), 6), np.float32)
            m = C.c_double()
            check(L.ecc_direct_evaluate_for_image_pair_kappas(self._h, int(i), int(j), len(kap), C.c_void_p(kap.ctypes.data),
                                                              C.c_void_p(s0.ctypes.data), C.c_void_p(s1.ctypes.data),
                                                              C.c_void_p(lines.ctypes.data), C.byref(m)))
            return m.value, dict(redundant_samples0=s0, redundant_samples1=s1, kappas=kap, lines=lines)
        cap = C.c_int()
        check(L.ecc_direct_lines_bound(self._h, C.byref(cap)))
        cap = cap.value
        s0, s1, kap = (np.empty(cap, np.float32) for _ in range(3))
        lines = np.empty((cap, 6), np.float32)
        n, m = C.c_int(), C.c_double()
        check(L.ecc_direct_evaluate_for_image_pair(self._h, int(i), int(j), cap, C.byref(n), C.c_void_p(s0.ctypes.data),
                                                   C.c_void_p(s1.ctypes.data), C.c_void_p(kap.ctypes.data),
                                                   C.c_void_p(lines.ctypes.data), C.byref(m)))
        n = n.value
        return m.value, dict(redundant_samples0=s0[:n], redundant_samples1=s1[:n], kappas=kap[:n], lines=lines[:n])

    def close(self):
        if self._h and self.ctx._h:
            _lib.lib().ecc_direct_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def slab_floats(n_alpha, n_t):
    """Floats per dtr in the private device layout (csrc/ecc_layout.h)."""
    return _lib.lib().ecc_dtr_slab_floats(int(n_alpha), int(n_t))


def get_ij(ij, n):
    i, j = C.c_int(), C.c_int()
    _lib.lib().ecc_get_ij(int(ij), int(n), C.byref(i), C.byref(j))
    return i.value, j.value


def host_pinvT(P):
    P = np.ascontiguousarray(np.asarray(P, np.float64).reshape(3, 4).T).reshape(12)
    out = np.empty(12, np.float32)
    _lib.lib().ecc_host_pinvT(C.c_void_p(P.ctypes.data), C.c_void_p(out.ctypes.data))
    return out


def host_source_position(P):
    P = np.ascontiguousarray(np.asarray(P, np.float64).reshape(3, 4).T).reshape(12)
    out = np.empty(4, np.float32)
    _lib.lib().ecc_host_source_position(C.c_void_p(P.ctypes.data), C.c_void_p(out.ctypes.data))
    return out


def host_object_radius(P, n_u, n_v):
    P = np.ascontiguousarray(np.asarray(P, np.float64).reshape(3, 4).T).reshape(12)
    return _lib.lib().ecc_host_object_radius(C.c_void_p(P.ctypes.data), int(n_u), int(n_v))


def host_intrinsics(P):
    """(sdd_px, ppu, ppv) = K(0,0), K(0,2), K(1,2) of P = K[R|t] (ref: getCameraIntrinsics)."""
    P = np.ascontiguousarray(np.asarray(P, np.float64).reshape(3, 4).T).reshape(12)
    a, b, c = C.c_float(), C.c_float(), C.c_float()
    _lib.lib().ecc_host_intrinsics(C.c_void_p(P.ctypes.data), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def _P12(P):
    return np.ascontiguousarray(np.asarray(P, np.float64).reshape(3, 4).T).reshape(12)


def estimateAngularRange(P0, P1, object_radius_mm):
    """ref: estimateAngularRange(join_pluecker(C0, C1), radius) (EpipolarConsistency.cpp:49-59)."""
    a, b, A, B = C.c_double(), C.c_double(), _P12(P0), _P12(P1)
    _lib.lib().ecc_host_angular_range(C.c_void_p(A.ctypes.data), C.c_void_p(B.ctypes.data), float(object_radius_mm),
                                      C.byref(a), C.byref(b))
    return a.value, b.value


def estimateAngularStep(P0, P1, n_u, n_v):
    """ref: estimateAngularStep (EpipolarConsistency.cpp:61-68)."""
    A, B = _P12(P0), _P12(P1)
    return _lib.lib().ecc_host_angular_step(C.c_void_p(A.ctypes.data), C.c_void_p(B.ctypes.data), int(n_u), int(n_v))


def estimateIsoCenter(Ps):
    """ref: estimateIsoCenter (EpipolarConsistency.cpp:8-33)."""
    flat = _Ps_colmajor(Ps)
    O = np.zeros(4)
    _lib.lib().ecc_host_iso_center(C.c_void_p(flat.ctypes.data), len(flat), C.c_void_p(O.ctypes.data))
    return O


estimateObjectRadius = host_object_radius  # ref: estimateObjectRadius (EpipolarConsistency.cpp:35-47)
