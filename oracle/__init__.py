"""ctypes doorway onto the CPU oracle (oracle/ecc_oracle.c) and, when built, onto the reference's
own headers (oracle/_ref/libecc_ref.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under epipolarconsistency_amd/ imports this package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(native=False):
    """(Re)build the oracle with oracle/Makefile.  Building the checker is not using it."""
    target = ["native"] if native else []
    subprocess.run(["make", "-s", "-C", _HERE] + target, check=True)


def _load(name, native=False):
    path = os.environ.get("ECC_ORACLE_LIB") or os.path.join(_HERE, name)  # override: the sanitizer build (scripts/sanitize.sh)
    if not os.path.exists(path):
        build(native=native)
    return C.CDLL(path)


_lib = None
_ref = None


def lib(native=False):
    """The oracle shared library (portable build, or -march=native for CPU-baseline timing)."""
    global _lib
    if native:
        return _bind(_load("libecc_oracle_native.so", native=True))
    if _lib is None:
        _lib = _bind(_load("libecc_oracle.so"))
    return _lib


def _bind(L):
    L.eccor_pinvT.argtypes = [_f64p, _f32p]
    L.eccor_source_position.argtypes = [_f64p, _f32p]
    L.eccor_camera_center.argtypes = [_f64p, _f64p]
    L.eccor_focal_length_px.argtypes = [_f64p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.eccor_object_radius.argtypes = [_f64p, C.c_int, C.c_int]
    L.eccor_object_radius.restype = C.c_double
    L.eccor_get_ij.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.eccor_computeK01.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_void_p, _f32p, _f32p,
                                   C.c_float, C.c_float, C.c_float, _f32p, _f32p]
    L.eccor_line_to_sample_dtr.argtypes = [_f32p, C.c_float]
    L.eccor_line_to_sample_dtr.restype = C.c_int
    L.eccor_tex2d.argtypes = [_f32p, C.c_int, C.c_int, C.c_float, C.c_float]
    L.eccor_tex2d.restype = C.c_float
    L.eccor_tex2d_norm.argtypes = [_f32p, C.c_int, C.c_int, C.c_float, C.c_float]
    L.eccor_tex2d_norm.restype = C.c_float
    L.eccor_radon.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32p,
                              C.POINTER(C.c_longlong)]
    L.eccor_ramp_kernel.argtypes = [C.c_int, _f64p]
    L.eccor_ramp_filter.argtypes = [_f32p, C.c_int, C.c_int]
    L.eccor_radon_bins.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   _i32p, C.c_int, _f32p]
    L.eccor_evaluate_all.argtypes = [C.c_int, _f64p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.POINTER(C.c_longlong)]
    L.eccor_evaluate_all.restype = C.c_double
    L.eccor_evaluate_pairs.argtypes = [C.c_int, _f64p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, _i32p,
                                       C.c_int, _f32p, C.c_void_p]
    L.eccor_evaluate_pairs.restype = C.c_double
    L.eccor_num_threads.restype = C.c_int
    L.eccor_evaluate_for_image_pair.argtypes = [_f64p, _f64p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, _f32p, _f32p,
                                                _f32p, _f32p, _f32p, _f32p, C.POINTER(C.c_double)]
    L.eccor_evaluate_for_image_pair.restype = C.c_int
    return L


def ref():
    """The reference's own headers behind extern "C" (oracle/ref_shim.cpp), or None if the
    prebuilt oracle/_ref/libecc_ref.so is absent and the reference tree is not present."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libecc_ref.so")
        if not os.path.exists(path):
            if not os.path.isdir("/root/reference/code"):
                return None
            build()
        R = C.CDLL(path)
        R.ref_pinvT.argtypes = [_f64p, _f32p]
        R.ref_source_position.argtypes = [_f64p, _f32p]
        R.ref_get_ij.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        R.ref_computeK01.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_void_p, _f32p, _f32p,
                                     C.c_float, C.c_float, C.c_float, _f32p, _f32p]
        R.ref_line_to_sample_dtr.argtypes = [_f32p, C.c_float]
        R.ref_line_to_sample_dtr.restype = C.c_int
        R.ref_weighting.argtypes = [C.c_float]
        R.ref_weighting.restype = C.c_float
        R.ref_gaussian_kernel.argtypes = [C.c_double, C.c_int, _f64p]
        R.ref_lowpass2D.argtypes = [_f32p, C.c_int, C.c_int, C.c_double, C.c_int]
        R.ref_image_view_at.argtypes = [_f32p, C.c_int, C.c_int, C.c_double, C.c_double]
        R.ref_image_view_at.restype = C.c_double
        _ref = R
    return _ref


# ---------------------------------------------------------------------------------------------
# numpy-level helpers
# ---------------------------------------------------------------------------------------------

def _P(P):
    """3x4 matrix (any layout numpy gives) -> 12 doubles column-major (Eigen default)."""
    P = np.asarray(P, dtype=np.float64).reshape(3, 4)
    return np.ascontiguousarray(P.T).reshape(12)


def pack_Ps(Ps):
    return np.ascontiguousarray(np.stack([_P(P) for P in Ps]).reshape(-1))


def pinvT(P, use_ref=False):
    out = np.zeros(12, np.float32)
    (ref().ref_pinvT if use_ref else lib().eccor_pinvT)(_P(P), out)
    return out


def source_position(P, use_ref=False):
    out = np.zeros(4, np.float32)
    (ref().ref_source_position if use_ref else lib().eccor_source_position)(_P(P), out)
    return out


def object_radius(P, n_u, n_v):
    return lib().eccor_object_radius(_P(P), int(n_u), int(n_v))


def get_ij(ij, n, use_ref=False):
    i, j = C.c_int(), C.c_int()
    (ref().ref_get_ij if use_ref else lib().eccor_get_ij)(ij, n, C.byref(i), C.byref(j))
    return i.value, j.value


def computeK01(n_x2, n_y2, C0, C1, P0invT, P1invT, object_radius_mm, num_samples, dkappa=0.0,
               use_ref=False):
    C0 = np.ascontiguousarray(C0, np.float32)
    C1 = np.ascontiguousarray(C1, np.float32)
    P0 = np.ascontiguousarray(P0invT, np.float32)
    P1 = np.ascontiguousarray(P1invT, np.float32)
    K0 = np.zeros(8, np.float32)
    K1 = np.zeros(8, np.float32)
    f = ref().ref_computeK01 if use_ref else lib().eccor_computeK01
    f(n_x2, n_y2, C0.ctypes.data, C1.ctypes.data, P0, P1, object_radius_mm, num_samples, dkappa,
      K0, K1)
    return K0, K1


def line_to_sample_dtr(line, range_t, use_ref=False):
    l = np.ascontiguousarray(line, np.float32).copy()
    f = ref().ref_line_to_sample_dtr if use_ref else lib().eccor_line_to_sample_dtr
    moved = f(l, range_t)
    return l, bool(moved)


class _radon_contract:
    """eccor_set_radon_contract around one call: the sampling loop of the Radon intermediate in contracted arithmetic
    (fmaf positions, 3-difference / 3-fmaf texel rule) -- the variant the product's ECC_RADON_FMA mode is pinned to."""

    def __init__(self, L, on):
        self.L, self.on = L, on

    def __enter__(self):
        self.L.eccor_set_radon_contract.argtypes = [C.c_int]
        self.L.eccor_set_radon_contract(1 if self.on else 0)

    def __exit__(self, *a):
        self.L.eccor_set_radon_contract(0)


def radon(img, n_alpha, n_t, filter=0, post=0, count_fetches=False, contract=False, native=False):
    """img: (n_v, n_u) float32.  Returns (n_t, n_alpha) float32 [, fetch count].  contract: see _radon_contract."""
    img = np.ascontiguousarray(img, np.float32)
    n_v, n_u = img.shape
    out = np.zeros((n_t, n_alpha), np.float32)
    nf = C.c_longlong(0)
    L = lib(native)
    with _radon_contract(L, contract):
        L.eccor_radon(img, n_u, n_v, n_alpha, n_t, filter, post, out, C.byref(nf))
    return (out, nf.value) if count_fetches else out


def ramp_kernel(n_t):
    """h2 (2*n_t doubles) of the circular convolution Filter::Ramp amounts to."""
    h2 = np.zeros(2 * n_t, np.float64)
    lib().eccor_ramp_kernel(int(n_t), h2)
    return h2


def ramp_filter(dtr):
    """dtr: (n_t, n_alpha) float32 -> ramp filtered copy (ref: RadonIntermediate.cu:186-237)."""
    out = np.ascontiguousarray(dtr, np.float32).copy()
    n_t, n_alpha = out.shape
    lib().eccor_ramp_filter(out, n_alpha, n_t)
    return out


def radon_bins(img, n_alpha, n_t, bins, filter=0, post=0, native=False, contract=False):
    img = np.ascontiguousarray(img, np.float32)
    n_v, n_u = img.shape
    bins = np.ascontiguousarray(bins, np.int32)
    out = np.zeros(len(bins), np.float32)
    L = lib(native)
    with _radon_contract(L, contract):
        L.eccor_radon_bins(img, n_u, n_v, n_alpha, n_t, filter, post, bins, len(bins), out)
    return out


def _dtr_ptrs(dtrs):
    keep = [np.ascontiguousarray(d, np.float32) for d in dtrs]
    arr = (C.c_void_p * len(keep))(*[d.ctypes.data for d in keep])
    return keep, arr


def evaluate_all(Ps, dtrs, n_u, n_v, object_radius_mm=0.0, dkappa=0.0, is_derivative=True,
                 cost=None, want_K01=False, native=False):
    """All-pairs ECC.  dtrs: list of (n_t, n_alpha) float32.  Returns dict(mean, pairs, cost, K01s,
    n_kappa)."""
    n = len(dtrs)
    n_t, n_alpha = dtrs[0].shape
    keep, arr = _dtr_ptrs(dtrs)
    n_pairs = n * (n - 1) // 2
    pairs = np.zeros(n_pairs, np.float32)
    if cost is None:
        cost = np.zeros((n, n), np.float32)
    cost = np.ascontiguousarray(cost, np.float32)
    K01s = np.zeros((n_pairs, 16), np.float32) if want_K01 else None
    nk = C.c_longlong(0)
    mean = lib(native).eccor_evaluate_all(
        n, pack_Ps(Ps), arr, n_u, n_v, n_alpha, n_t, float(object_radius_mm), float(dkappa),
        1 if is_derivative else 0, cost.ctypes.data, pairs.ctypes.data,
        K01s.ctypes.data if want_K01 else None, C.byref(nk))
    return dict(mean=mean, pairs=pairs, cost=cost, K01s=K01s, n_kappa=nk.value)


def evaluate_pairs(Ps, dtrs, n_u, n_v, idx4, object_radius_mm=0.0, dkappa=0.0, is_derivative=True,
                   native=False):
    n_t, n_alpha = dtrs[0].shape
    keep, arr = _dtr_ptrs(dtrs)
    idx4 = np.ascontiguousarray(idx4, np.int32).reshape(-1, 4)
    out = np.zeros(len(idx4), np.float32)
    mean = lib(native).eccor_evaluate_pairs(
        len(Ps), pack_Ps(Ps), len(dtrs), arr, n_u, n_v, n_alpha, n_t, float(object_radius_mm),
        float(dkappa), 1 if is_derivative else 0, idx4.reshape(-1), len(idx4), out, None)
    return dict(mean=mean, pairs=out)


def evaluate_for_image_pair(Ps, dtrs, i, j, n_u, n_v, object_radius_mm=0.0, dkappa=0.0, derivative=True):
    """E7 (ref: ...RadonIntermediate.cpp:324-393, evident intent -- see ecc_oracle.c).  Returns dict(ecc, samples0,
    samples1, kappas, radon0 (n,2), radon1 (n,2), K01)."""
    n_t, n_alpha = dtrs[i].shape
    radius = float(object_radius_mm) if object_radius_mm > 0 else object_radius(Ps[0], n_u, n_v)
    cap = int(np.sqrt(float(n_u * n_u + n_v * n_v))) + 16 if dkappa <= 0 else int(np.pi / dkappa) + 16
    rs0, rs1, kap = (np.zeros(cap, np.float32) for _ in range(3))
    r0, r1 = np.zeros(2 * cap, np.float32), np.zeros(2 * cap, np.float32)
    K01 = np.zeros(16, np.float32)
    ecc = C.c_double()
    d0 = np.ascontiguousarray(dtrs[i], np.float32)
    d1 = np.ascontiguousarray(dtrs[j], np.float32)
    n = lib().eccor_evaluate_for_image_pair(_P(Ps[i]), _P(Ps[j]), d0, d1, n_u, n_v, n_alpha, n_t, radius, float(dkappa),
                                            1 if derivative else 0, 1 if derivative else 0, cap, rs0, rs1, kap, r0, r1,
                                            K01, C.byref(ecc))
    assert n <= cap
    return dict(ecc=ecc.value, samples0=rs0[:n], samples1=rs1[:n], kappas=kap[:n], radon0=r0[:2 * n].reshape(n, 2),
                radon1=r1[:2 * n].reshape(n, 2), K01=K01)


class PreprocessParams(C.Structure):
    """Mirror of eccor_preprocess_params (defaults = the reference's, Gui/PreProccess.h:19-45)."""
    _fields_ = [("process", C.c_int), ("normalize", C.c_int), ("bias", C.c_double), ("scale", C.c_double),
                ("apply_log", C.c_int), ("gaussian_sigma", C.c_double), ("half_kernel_width", C.c_int),
                ("flip_u", C.c_int), ("flip_v", C.c_int), ("zero", C.c_int * 4), ("feather", C.c_int * 4),
                ("n_blanks", C.c_int), ("blanks", C.c_void_p)]


def preprocess_params(normalize=False, bias=0.0, scale=1.0, apply_log=False, gaussian_sigma=1.84,
                      half_kernel_width=5, flip_u=False, flip_v=False, zero=(1, 1, 1, 1), feather=(16, 16, 16, 16),
                      blanks=(), process=True):
    p = PreprocessParams()
    p.process, p.normalize, p.bias, p.scale, p.apply_log = int(process), int(normalize), bias, scale, int(apply_log)
    p.gaussian_sigma, p.half_kernel_width, p.flip_u, p.flip_v = gaussian_sigma, half_kernel_width, int(flip_u), int(flip_v)
    p.zero = (C.c_int * 4)(*zero)
    p.feather = (C.c_int * 4)(*feather)
    bl = np.ascontiguousarray(np.asarray(blanks, np.int32).reshape(-1, 4))
    p.n_blanks = len(bl)
    p._keep = bl
    p.blanks = bl.ctypes.data if len(bl) else None
    return p


def preprocess(img, P=None, **kw):
    """ref: PreProccess::process + apply_weight_cos_principal_ray (Gui/PreProccess.cpp:57-166); returns a copy."""
    L = lib()
    L.eccor_preprocess.argtypes = [_f32p, C.c_int, C.c_int, C.POINTER(PreprocessParams)]
    L.eccor_cos_weight.argtypes = [_f32p, C.c_int, C.c_int, _f64p]
    out = np.ascontiguousarray(img, np.float32).copy()
    n_v, n_u = out.shape
    p = preprocess_params(**kw)
    L.eccor_preprocess(out, n_u, n_v, C.byref(p))
    if P is not None:
        L.eccor_cos_weight(out, n_u, n_v, _P(P))
    return out


def gaussian_kernel(sigma, k):
    """eccor_gaussian_kernel (ref: HeaderOnly/NRRD/nrrd_lowpass.hxx:19-34): 2k + 1 float64 taps."""
    L = lib()
    L.eccor_gaussian_kernel.argtypes = [C.c_double, C.c_int, _f64p]
    out = np.zeros(2 * k + 1, np.float64)
    L.eccor_gaussian_kernel(float(sigma), int(k), out)
    return out


def tex2d(img, x, y):
    """eccor_tex2d: the normative un-normalised bilinear rule."""
    L = lib()
    L.eccor_tex2d.argtypes = [_f32p, C.c_int, C.c_int, C.c_float, C.c_float]
    L.eccor_tex2d.restype = C.c_float
    img = np.ascontiguousarray(img, np.float32)
    return L.eccor_tex2d(img, img.shape[1], img.shape[0], float(x), float(y))


def intrinsics(P):
    L = lib()
    L.eccor_intrinsics.argtypes = [_f64p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    a, b, c = C.c_float(), C.c_float(), C.c_float()
    L.eccor_intrinsics(_P(P), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def direct_pair(P0, P1, img0, img1, dkappa=0.0, object_radius_mm=0.0, fbcc=False):
    """MetricDirect for one pair (ref: EpipolarConsistencyDirect.cpp:67-219; fbcc: the rectified fan-beam
    weighting of :133-196 instead of the derivative).  Returns dict(metric, samples0, samples1, kappas, lines (n,6))."""
    L = lib()
    L.eccor_set_direct_fbcc.argtypes = [C.c_int]
    L.eccor_set_direct_fbcc(1 if fbcc else 0)
    L.eccor_direct_pair.argtypes = [_f64p, _f64p, _f32p, _f32p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                    _f32p, _f32p, _f32p, _f32p, C.POINTER(C.c_double)]
    L.eccor_direct_pair.restype = C.c_int
    a = np.ascontiguousarray(img0, np.float32)
    b = np.ascontiguousarray(img1, np.float32)
    n_v, n_u = a.shape
    cap = (int(2 * np.sqrt(float(n_u * n_u + n_v * n_v))) if dkappa <= 0 else int(np.pi / dkappa)) + 16
    v0, v1, kap = (np.zeros(cap, np.float32) for _ in range(3))
    lines = np.zeros(6 * cap, np.float32)
    m = C.c_double()
    n = L.eccor_direct_pair(_P(P0), _P(P1), a, b, n_u, n_v, float(dkappa), float(object_radius_mm), cap, v0, v1, kap,
                            lines, C.byref(m))
    assert n <= cap
    return dict(metric=m.value, samples0=v0[:n], samples1=v1[:n], kappas=kap[:n], lines=lines[:6 * n].reshape(n, 6))


def direct_evaluate(Ps, imgs, dkappa=0.0, object_radius_mm=0.0, fbcc=False):
    """ref: MetricDirect::evaluate (EpipolarConsistencyDirect.cpp:247-259): SUM over pairs and the cost image
    (index i + j*n, i<j); the radius default is Metric::getObjectRadius (first projection)."""
    n = len(imgs)
    n_v, n_u = np.asarray(imgs[0]).shape
    radius = object_radius_mm if object_radius_mm > 0 else object_radius(Ps[0], n_u, n_v)
    cost = np.zeros((n, n), np.float32)
    total = 0.0
    for i in range(n):
        for j in range(i + 1, n):
            e = direct_pair(Ps[i], Ps[j], imgs[i], imgs[j], dkappa, radius, fbcc)["metric"]
            total += e
            cost[j, i] = e
    return dict(sum=total, cost=cost)


def set_variant(v, native=False):
    """0 = normative fp32 path (correctly rounded elementary functions); 1 = line->(angle,distance)
    mapping in binary64 (noise-floor probe); 2 = platform float libm (what oracle/_ref is built on)."""
    L = lib(native)
    L.eccor_set_variant.argtypes = [C.c_int]
    L.eccor_set_variant(int(v))


def set_use_corr(v):
    """MetricRadonIntermediate::useCorrelation: pair value = 1 - un-centred correlation of the two
    redundant signals (ref: ...RadonIntermediate.cu:116-149, .cpp:127-131,199-210).  Parity unpinned."""
    L = lib()
    L.eccor_set_use_corr.argtypes = [C.c_int]
    L.eccor_set_use_corr(1 if v else 0)
