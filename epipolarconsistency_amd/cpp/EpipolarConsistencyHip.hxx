// EpipolarConsistencyHip.hxx -- header-only C++ adapter: the reference's class names and method
// signatures for the hot path, forwarding to the C ABI of libecc_hip.so (include/ecc_hip.h).
//
//   EpipolarConsistency::RadonIntermediate        ref: LibEpipolarConsistency/RadonIntermediate.h:18-128
//   EpipolarConsistency::Metric                   ref: LibEpipolarConsistency/EpipolarConsistency.h:49-94
//   EpipolarConsistency::MetricRadonIntermediate  ref: LibEpipolarConsistency/EpipolarConsistencyRadonIntermediate.h:21-106
//   EpipolarConsistency::MetricDirect             ref: LibEpipolarConsistency/EpipolarConsistencyDirect.h:28-60
//   EpipolarConsistency::PreProccess              ref: LibEpipolarConsistency/Gui/PreProccess.h:13-50 (fields and the two image calls)
//   estimateIsoCenter / estimateObjectRadius / estimateAngularRange / estimateAngularStep
//                                                 ref: LibEpipolarConsistency/EpipolarConsistency.h:36-46
//
// A caller such as Gui/SingleImageMotion.h (:37,72,88), Gui/Registration.h (:34,67,80) or
// tools/Registration/Registration3D3D.hxx (:66-67,95) compiles against this header instead of the
// three reference headers above and links libecc_hip.so instead of LibEpipolarConsistency +
// LibUtilsCuda; nothing else changes.  With Eigen on the include path Geometry::ProjectionMatrix is
// the reference's Eigen::Matrix<double,3,4>; without it a 12-double column-major stand-in with the
// same .data() contract is used (that is what this repository's tests compile, Eigen is not
// installed here).  Images are taken as (pointer, width, height); with the reference's header-only NRRD
// library on the include path (-I<reference>/code/HeaderOnly) the reference's own NRRD-typed signatures are
// there as well: RadonIntermediate(const NRRD::ImageView<float>&, ...), RadonIntermediate(const std::string path),
// RadonIntermediate(const NRRD::ImageView<float>&), readPropertiesFromMeta, replaceRadonIntermediateData(view),
// and data() returns NRRD::ImageView<float>& (ref: RadonIntermediate.h:31-41,47,53,80-83); compiled by
// tests/test_cpp_adapter.py against the reference headers where they lie.
//
// Multi-GPU without touching the callers: a process-wide default group (setDefaultDevices(), or the environment
// variable ECC_HIP_DEVICES = "0,1,2,3" / "all") makes every MetricRadonIntermediate shard its all-pairs evaluate()
// over those devices (ecc_group_* of the C ABI); index-list / subset / single-pair calls stay on the first device.
//
// Error behaviour: the reference prints and exit()s on any CUDA error (LibUtilsCuda/UtilsCuda.hxx:14-28);
// the adapter throws std::runtime_error with ecc_last_error() instead.
// Not carried over: BindlessTexture2D (there are no textures; getTexture() returns null),
// setProjectionImages (a stub in the reference, ...RadonIntermediate.cpp:121-125).
#ifndef ECC_EPIPOLAR_CONSISTENCY_HIP_HXX
#define ECC_EPIPOLAR_CONSISTENCY_HIP_HXX

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "ecc_hip.h"

#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define ECC_ADAPTER_HAVE_EIGEN 1
#endif
#if __has_include(<NRRD/nrrd_image.hxx>)
#include <NRRD/nrrd_image.hxx>  // the reference's header-only NRRD library (ImageView, Image, load/save)
#define ECC_ADAPTER_HAVE_NRRD 1
#endif
#endif

namespace Geometry {
#ifdef ECC_ADAPTER_HAVE_EIGEN
typedef Eigen::Matrix<double, 3, 4> ProjectionMatrix;  // ref: LibProjectiveGeometry/ProjectiveGeometry.hxx:22
#else
/// 3x4, column-major like Eigen's default: element (r, c) at data()[r + 3*c].
struct ProjectionMatrix {
    double v[12];
    ProjectionMatrix() { for (double& x : v) x = 0; }
    double& operator()(int r, int c) { return v[r + 3 * c]; }
    double operator()(int r, int c) const { return v[r + 3 * c]; }
    const double* data() const { return v; }
    double* data() { return v; }
};
#endif
}  // namespace Geometry

/// Projection tables as text, one matrix per line ("[a b c d; e f g h; i j k l]"), '#' comments, "#> key="value" ..."
/// attribute lines -- the .ompl files the reference's tools exchange
/// (ref: HeaderOnly/Utils/Projtable.hxx:168-220, LibProjectiveGeometry/EigenToStr.hxx).
namespace ProjTable {

/// ref: loadProjectionsOneMatrixPerLine (Projtable.hxx:168-188): '#>' lines fill *meta, the first other comment is
/// stored under "comment", every other non-empty line is a matrix.
inline std::vector<Geometry::ProjectionMatrix> loadProjectionsOneMatrixPerLine(const std::string& file,
                                                                               std::map<std::string, std::string>* meta = 0x0)
{
    std::vector<Geometry::ProjectionMatrix> ret;
    std::ifstream pt(file.c_str());
    std::string line;
    while (pt && std::getline(pt, line)) {
        while (!line.empty() && (line[line.size() - 1] == '\r' || line[line.size() - 1] == '\n')) line.erase(line.size() - 1);
        if (line.empty()) continue;
        if (line[0] == '#') {
            if (meta && line.size() > 1 && line[1] == '>') {
                // key="value" pairs
                size_t pos = 2;
                while (pos < line.size()) {
                    const size_t eq = line.find("=\"", pos);
                    if (eq == std::string::npos) break;
                    const size_t end = line.find('"', eq + 2);
                    if (end == std::string::npos) break;
                    size_t k0 = line.find_first_not_of(" \t", pos);
                    (*meta)[line.substr(k0, eq - k0)] = line.substr(eq + 2, end - eq - 2);
                    pos = end + 1;
                }
            } else if (meta && meta->find("comment") == meta->end())
                (*meta)["comment"] = line.substr(1);
            continue;
        }
        for (size_t i = 0; i < line.size(); ++i)
            if (line[i] == '[' || line[i] == ']' || line[i] == ';' || line[i] == ',') line[i] = ' ';
        std::istringstream in(line);
        Geometry::ProjectionMatrix P;
        bool ok = true;
        for (int r = 0; r < 3 && ok; ++r)
            for (int c = 0; c < 4 && ok; ++c) {
                double v = 0;
                ok = !!(in >> v);
                P(r, c) = v;
            }
        if (ok) ret.push_back(P);
    }
    return ret;
}

/// ref: saveProjectionsOneMatrixPerLine (Projtable.hxx:199-220); detector_size_px: null = not written.
inline bool saveProjectionsOneMatrixPerLine(const std::vector<Geometry::ProjectionMatrix>& Ps, const std::string& path,
                                            const std::string& first_line_comment = "", double spacing = 0.0,
                                            const int* detector_size_px = 0x0)
{
    std::ofstream file(path.c_str());
    if (!file) return false;
    file.precision(17);  // max_digits10: a saved table loads back bit for bit (the reference writes 12 digits; readers take either)
    if (!first_line_comment.empty()) file << "#" << first_line_comment << std::endl;
    if (spacing != 0.0) {
        file << "#> spacing=\"" << spacing << "\"";
        if (detector_size_px) file << " detector_size_px=\"" << detector_size_px[0] << " " << detector_size_px[1] << "\"";
        file << std::endl;
    }
    for (size_t i = 0; i < Ps.size(); ++i) {
        file << "[";
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 4; ++c) file << (c ? " " : "") << Ps[i](r, c);
            file << (r < 2 ? "; " : "] ");
        }
        file << std::endl;
    }
    return true;
}

}  // namespace ProjTable

namespace EpipolarConsistency {

using Geometry::ProjectionMatrix;

namespace detail {
inline void check(int rc)
{
    if (rc != ECC_OK) throw std::runtime_error(std::string("libecc_hip: ") + ecc_last_error());
}
/// Process-wide default group (null: single device).  Set by setDefaultDevices() or, on first use, from the
/// environment variable ECC_HIP_DEVICES ("0,1,2,3" or "all").
inline ecc_group*& default_group_slot()
{
    static ecc_group* g = nullptr;
    return g;
}
inline ecc_group* default_group()
{
    static bool env_checked = false;
    ecc_group*& g = default_group_slot();
    if (!g && !env_checked) {
        env_checked = true;
        const char* e = std::getenv("ECC_HIP_DEVICES");
        if (e && *e) {
            std::vector<int> devs;
            if (std::strcmp(e, "all") == 0) {
                for (int d = 0; d < ecc_device_count(); ++d) devs.push_back(d);
            } else {
                for (const char* p = e; *p;) {
                    devs.push_back(std::atoi(p));
                    while (*p && *p != ',') ++p;
                    if (*p == ',') ++p;
                }
            }
            if (devs.size() > 1) check(ecc_group_create((int)devs.size(), devs.data(), &g));
        }
    }
    return g;
}
/// One context per (device, stream); the reference uses the current device's default stream.  With a default group
/// this is the group's first context, so that dtrs and the group's rank 0 share a device and a stream.
inline ecc_ctx* default_context()
{
    static ecc_ctx* ctx = nullptr;
    if (!ctx) {
        if (ecc_group* g = default_group()) check(ecc_group_ctx(g, 0, &ctx));
        else check(ecc_ctx_create(0, nullptr, &ctx));
    }
    return ctx;
}
}  // namespace detail

/// Not in the reference: ecc_radon_set_arithmetic of the default context -- RadonIntermediates computed afterwards use the
/// exact (default; unfused fp32, the CPU reading of ref: RadonIntermediate.cu:118-123) or the contracted convention
/// (positions fmaf(t, d, o), lerps as one fma each: the arithmetic class of the reference's GPU build; ~11 % faster, the
/// ECC metric moves by < 2e-6 relative on means over pairs).
inline void setRadonArithmetic(bool contracted) { detail::check(ecc_radon_set_arithmetic(detail::default_context(), contracted ? ECC_RADON_FMA : ECC_RADON_EXACT)); }

/// Not in the reference: ecc_ctx_set_quad_copies of the default context -- whether MetricRadonIntermediates created afterwards
/// keep row-quad copies of their Radon intermediates (4x the slab memory; the pairs whose baseline passes through the object
/// sample them; bit-identical values, ~2 % per evaluation).  Default: ECC_QUAD_COPIES_AUTO (built while they fit).
inline void setQuadCopies(int mode) { detail::check(ecc_ctx_set_quad_copies(detail::default_context(), mode)); }

/// Not in the reference: evaluate() of every MetricRadonIntermediate created afterwards is sharded over these HIP
/// devices (one host thread per device inside the library).  Call before the first dtr / metric is created.
inline void setDefaultDevices(const std::vector<int>& devices)
{
    ecc_group*& g = detail::default_group_slot();
    if (g) throw std::runtime_error("setDefaultDevices: a default group exists already");
    if (devices.size() > 1) detail::check(ecc_group_create((int)devices.size(), devices.data(), &g));
}

/// ref: class RadonIntermediate
class RadonIntermediate {
public:
    enum Filter { Derivative = 0, Ramp = 1, None = 2 };
    enum PostProcess { Identity = 0, SquareRoot = 1, Logarithm = 2 };

    /// ref: RadonIntermediate(const NRRD::ImageView<float>&, size_alpha, size_t, filter, post_process)
    RadonIntermediate(const float* projectionData, int n_u, int n_v, int size_alpha, int size_t, Filter filter,
                      PostProcess post_process, ecc_ctx* ctx = nullptr)
        : m_h(nullptr)
    {
        detail::check(ecc_radon_compute(ctx ? ctx : detail::default_context(), projectionData, 0, n_u, n_v, size_alpha,
                                        size_t, (int)filter, (int)post_process, &m_h));
    }
#ifdef ECC_ADAPTER_HAVE_NRRD
    /// ref: RadonIntermediate(const NRRD::ImageView<float>& projectionData, size_alpha, size_t, filter, post_process)
    /// (RadonIntermediate.h:31, .cpp:17-31)
    RadonIntermediate(const NRRD::ImageView<float>& projectionData, int size_alpha, int size_t, Filter filter,
                      PostProcess post_process)
        : RadonIntermediate((const float*)projectionData, projectionData.size(0), projectionData.size(1), size_alpha,
                            size_t, filter, post_process)
    {
    }
    /// ref: RadonIntermediate(const std::string path) (RadonIntermediate.h:37, .cpp:47-67): a dtr NRRD file written
    /// by the reference's tools or by save(); size and filter come from its meta info (readPropertiesFromMeta).
    /// Like the reference, a file that cannot be read leaves an invalid object (message on stderr).
    explicit RadonIntermediate(const std::string path) : m_h(nullptr)
    {
        if (!m_raw_cpu.load(path) || !m_raw_cpu) {
            std::cerr << "Failed to load " << path << std::endl;
            return;
        }
        upload_raw_cpu(m_raw_cpu.meta_info);
    }
    /// ref: RadonIntermediate(const NRRD::ImageView<float>& radon_intermediate_image) (RadonIntermediate.h:40,
    /// .cpp:69-80): existing dtr data in host memory, size(0) = angle bins (fast), size(1) = distance bins; original
    /// image size and filter from its meta_info.
    explicit RadonIntermediate(const NRRD::ImageView<float>& radon_intermediate_image) : m_h(nullptr)
    {
        replaceRadonIntermediateData(radon_intermediate_image);
    }
    /// ref: readPropertiesFromMeta (RadonIntermediate.h:47, .cpp:82-93).  The bin sizes are functions of the bin
    /// counts and the original image size here (ecc_dtr_info), so only size and filter are taken from the dictionary;
    /// the device copy is re-created with them.
    void readPropertiesFromMeta(std::map<std::string, std::string> dict)
    {
        if (m_raw_cpu.size(0) < 1) readback();
        upload_raw_cpu(dict);
    }
    /// ref: replaceRadonIntermediateData(const NRRD::ImageView<float>&) (RadonIntermediate.h:53, .cpp:105-123)
    void replaceRadonIntermediateData(const NRRD::ImageView<float>& radon_intermediate_image)
    {
        m_raw_cpu.clone(radon_intermediate_image);
        upload_raw_cpu(radon_intermediate_image.meta_info);
    }
    /// ref: data() (RadonIntermediate.h:80-83): the host copy made by readback() (may be an invalid image).
    NRRD::ImageView<float>& data() { return m_raw_cpu; }
    const NRRD::ImageView<float>& data() const { return m_raw_cpu; }
    /// Save like the reference's tools do (Gui/ComputeRadonIntermediate.hxx:77-83): readback, meta info, NRRD file.
    bool save(const std::string& path)
    {
        readback();
        writePropertiesToMeta(m_raw_cpu.meta_info);
        return m_raw_cpu.save(path);
    }
#endif
    /// ref: RadonIntermediate(const NRRD::ImageView<float>& radon_intermediate_image): existing dtr data,
    /// n_t rows x n_alpha columns (alpha fastest); the original image size comes from the meta info there.
    RadonIntermediate(const float* radon_intermediate_image, int n_alpha, int n_t, int original_n_u, int original_n_v,
                      Filter filter, ecc_ctx* ctx = nullptr)
        : m_h(nullptr)
    {
        detail::check(ecc_dtr_from_host(ctx ? ctx : detail::default_context(), radon_intermediate_image, n_alpha, n_t,
                                        original_n_u, original_n_v, (int)filter, &m_h));
    }
    /// Adopts a handle produced by ecc_radon_compute_batch.
    explicit RadonIntermediate(ecc_dtr* handle) : m_h(handle) {}
    ~RadonIntermediate() { ecc_dtr_destroy(m_h); }

    Filter getFilter() const { return (Filter)info().filter; }
    bool isDerivative() const { return getFilter() == Derivative; }
    /// 0: angle bins, 1: distance bins (ref: RadonIntermediate.cpp:165-168)
    int getRadonBinNumber(int dim) const { return dim ? info().n_t : info().n_alpha; }
    int getOriginalImageSize(int dim) const { return dim ? info().n_v : info().n_u; }
    /// 0: angle, 1: distance (ref: RadonIntermediate.cpp:175-178)
    double getRadonBinSize(int dim = 1) const { return dim ? info().bin_distance : info().bin_angle; }

    /// ref: readback() + data(): host copy, n_t rows x n_alpha columns, alpha fastest.
    void readback(bool /*gpu_memory_only*/ = false)
    {
        Info i = info();
        host_resize(i.n_alpha, i.n_t);
        detail::check(ecc_dtr_readback(m_h, host_ptr()));
    }
#ifndef ECC_ADAPTER_HAVE_NRRD
    const std::vector<float>& data() const { return m_raw_cpu; }
#endif
    void clearRawData() { host_resize(0, 0); }
    /// ref: getTexture() (RadonIntermediate.h:76, .cpp:188-196) uploads host data that is not on the device yet and
    /// returns the bindless texture.  Here the device copy is always current and there are no textures: returns null.
    /// Kept because a caller uses it for the upload alone (Gui/InputDataRadonIntermediate.cpp:78).
    void* getTexture() { return nullptr; }

    /// ref: replaceRadonIntermediateData(image) (RadonIntermediate.cpp:105-123): new data from host memory, n_t rows x
    /// n_alpha columns; original image size and filter are kept.  A metric that already holds this object must be
    /// given the dtrs again (setRadonIntermediates), like in the reference.
    void replaceRadonIntermediateData(const float* radon_intermediate_image, int n_alpha, int n_t, ecc_ctx* ctx = nullptr)
    {
        Info i = info();
        ecc_dtr* fresh = nullptr;
        detail::check(ecc_dtr_from_host(ctx ? ctx : detail::default_context(), radon_intermediate_image, n_alpha, n_t, i.n_u,
                                        i.n_v, i.filter, &fresh));
        ecc_dtr_destroy(m_h);
        m_h = fresh;
        host_resize(n_alpha, n_t);
        std::memcpy(host_ptr(), radon_intermediate_image, sizeof(float) * (size_t)n_alpha * n_t);
    }

    /// ref: tex2D(s, t) (RadonIntermediate.h:108): host sample of the read-back data in texture coordinates [0,1]^2,
    /// bilinear in binary64 on the (n - 1)-scaled grid with NRRD::ImageView's edge rule
    /// (HeaderOnly/NRRD/nrrd_image_view.hxx:159-205).  readback() first.
    float tex2D(float s, float t) const
    {
        Info i = info();
        if (host_length() != (size_t)i.n_alpha * i.n_t) throw std::runtime_error("RadonIntermediate::tex2D: call readback() first");
#ifdef ECC_ADAPTER_HAVE_NRRD
        return (float)m_raw_cpu((i.n_alpha - 1) * s, (i.n_t - 1) * t);  // the reference's expression (RadonIntermediate.h:108)
#endif
        double x = (i.n_alpha - 1) * (double)s, y = (i.n_t - 1) * (double)t;
        int ix = (int)x, iy = (int)y;
        double fx = x - ix, fy = y - iy;
        if (ix < 0) { ix = 0; fx = 0; }
        if (ix > i.n_alpha - 2) { ix = i.n_alpha - 2; fx = 1.0; }
        if (iy < 0) { iy = 0; fy = 0; }
        if (iy > i.n_t - 2) { iy = i.n_t - 2; fy = 1.0; }
        const float* I = host_ptr();
        const size_t w = (size_t)i.n_alpha;
        if (fx == 0 && fy == 0) return I[ix + iy * w];
        return (float)((1.0 - fy) * ((1.0 - fx) * I[ix + iy * w] + fx * I[ix + 1 + iy * w]) +
                       fy * ((1.0 - fx) * I[ix + (iy + 1) * w] + fx * I[ix + 1 + (iy + 1) * w]));
    }

    /// ref: sample(line) (RadonIntermediate.h:86-105): host sample for a line (l0, l1, l2) relative to the image centre;
    /// `line` is overwritten with the sample location.  Negated on the folded branch of a derivative dtr -- the evident
    /// intent: the reference's own flip test comes after lineToSampleDtr has folded the angle and can never fire.
    template <typename Line>
    float sample(Line& line) const
    {
        float l[3] = {(float)line[0], (float)line[1], (float)line[2]};
        const float range_t = (float)getRadonBinSize(1) * getRadonBinNumber(1);
        const int folded = ecc_host_line_to_sample_dtr(l, range_t);
        line[0] = l[0];
        line[1] = l[1];
        const float v = tex2D(l[0], l[1]);
        return (folded && isDerivative()) ? -v : v;
    }

    /// ref: writePropertiesToMeta (RadonIntermediate.cpp:95-103)
    void writePropertiesToMeta(std::map<std::string, std::string>& dict) const
    {
        Info i = info();
        dict["Bin Size/Angle"] = std::to_string(i.bin_angle);
        dict["Bin Size/Distance"] = std::to_string(i.bin_distance);
        dict["Original Image/Width"] = std::to_string(i.n_u);
        dict["Original Image/Height"] = std::to_string(i.n_v);
        dict["Filter"] = i.filter == Derivative ? "Derivative" : (i.filter == Ramp ? "Ramp" : "None");
    }

    ecc_dtr* handle() const { return m_h; }

private:
    RadonIntermediate(const RadonIntermediate&);
    RadonIntermediate& operator=(const RadonIntermediate&);
    struct Info {
        int n_alpha, n_t, n_u, n_v, filter;
        double bin_angle, bin_distance;
    };
    Info info() const
    {
        Info i;
        detail::check(ecc_dtr_info(m_h, &i.n_alpha, &i.n_t, &i.n_u, &i.n_v, &i.filter, &i.bin_angle, &i.bin_distance));
        return i;
    }
    ecc_dtr* m_h;
#ifdef ECC_ADAPTER_HAVE_NRRD
    NRRD::Image<float> m_raw_cpu;  // ref: RadonIntermediate.h:111
    void host_resize(int n_alpha, int n_t)
    {
        if (n_alpha < 1 || n_t < 1) m_raw_cpu.set(0, 0, 0);
        else if (m_raw_cpu.size(0) != n_alpha || m_raw_cpu.size(1) != n_t) m_raw_cpu.set(n_alpha, n_t);
    }
    float* host_ptr() const { return (float*)m_raw_cpu; }
    size_t host_length() const { return !m_raw_cpu ? 0 : (size_t)m_raw_cpu.length(); }
    /// (re)creates the device copy from m_raw_cpu with size / filter taken from a meta dictionary
    void upload_raw_cpu(std::map<std::string, std::string> dict)
    {
        const int n_x = stringTo<int>(dict["Original Image/Width"]), n_y = stringTo<int>(dict["Original Image/Height"]);
        const Filter f = dict["Filter"] == "Ramp" ? Ramp : (dict["Filter"] == "Derivative" ? Derivative : None);
        ecc_dtr* fresh = nullptr;
        detail::check(ecc_dtr_from_host(detail::default_context(), host_ptr(), m_raw_cpu.size(0), m_raw_cpu.size(1), n_x, n_y,
                                        (int)f, &fresh));
        ecc_dtr_destroy(m_h);
        m_h = fresh;
    }
#else
    std::vector<float> m_raw_cpu;
    void host_resize(int n_alpha, int n_t) { std::vector<float>((size_t)(n_alpha > 0 ? n_alpha : 0) * (n_t > 0 ? n_t : 0)).swap(m_raw_cpu); }
    float* host_ptr() const { return const_cast<float*>(m_raw_cpu.data()); }
    size_t host_length() const { return m_raw_cpu.size(); }
#endif
};

/// ref: struct PreProccess (Gui/PreProccess.h:13-50) without the GetSet GUI glue: the same fields and defaults; the two
/// image calls run as one device kernel each (ecc_preprocess of the C ABI, bit-identical to the reference's host loops).
/// The reference's order is process(image); apply_weight_cos_principal_ray(image, P) (Gui/InputDataDirect.cpp:85-86);
/// process_and_weight does both in ONE pass over a whole stack.
struct PreProccess {
#ifdef ECC_ADAPTER_HAVE_EIGEN
    typedef Eigen::Vector4i Vector4i;
    static Vector4i constant4(int v) { return Vector4i::Constant(v); }
#else
    struct Vector4i {
        int v[4];
        int& operator[](int i) { return v[i]; }
        int operator[](int i) const { return v[i]; }
    };
    static Vector4i constant4(int x) { Vector4i r; r.v[0] = r.v[1] = r.v[2] = r.v[3] = x; return r; }
#endif
    struct Intensity {
        bool normalize; double bias; double scale; bool apply_log;
        Intensity() : normalize(false), bias(0.0), scale(1.0), apply_log(false) {}
    } intensity;
    struct Lowpass {
        double gaussian_sigma; int half_kernel_width;
        Lowpass() : gaussian_sigma(1.84), half_kernel_width(5) {}
    } lowpass;
    struct ImageGeometry {
        bool flip_u, flip_v;
        ImageGeometry() : flip_u(false), flip_v(false) {}
    } image_geometry;
    /// Offsets are left, right, bottom, top; blanks are rectangles (x0, y0, x1, y1).
    struct Border {
        Vector4i zero, feather;
        std::vector<Vector4i> blanks;
        Border() : zero(constant4(1)), feather(constant4(16)) {}
    } border;

    /// ref: process(NRRD::ImageView<float>& image) (Gui/PreProccess.cpp:57-144), in place on a host image
    void process(float* image, int n_u, int n_v, ecc_ctx* ctx = nullptr) const { run(image, 1, n_u, n_v, true, nullptr, ctx); }
    /// ref: apply_weight_cos_principal_ray(image, P) (Gui/PreProccess.cpp:146-166), in place on a host image
    void apply_weight_cos_principal_ray(float* image, int n_u, int n_v, const ProjectionMatrix& P, ecc_ctx* ctx = nullptr) const
    {
        run(image, 1, n_u, n_v, false, P.data(), ctx);
    }
    /// Both steps for a stack of n host images (n x 12 doubles of matrices), one kernel launch.
    void process_and_weight(float* images, int n, int n_u, int n_v, const std::vector<ProjectionMatrix>& Ps, ecc_ctx* ctx = nullptr) const
    {
        std::vector<double> flat(12 * Ps.size());
        for (size_t i = 0; i < Ps.size(); ++i)
            for (int k = 0; k < 12; ++k) flat[12 * i + k] = Ps[i].data()[k];
        if ((int)Ps.size() != n) throw std::runtime_error("PreProccess::process_and_weight: one matrix per image");
        run(images, n, n_u, n_v, true, flat.data(), ctx);
    }
#ifdef ECC_ADAPTER_HAVE_NRRD
    void process(NRRD::ImageView<float>& image) const { process((float*)image, image.size(0), image.size(1)); }
    void apply_weight_cos_principal_ray(NRRD::ImageView<float>& image, const ProjectionMatrix& P) const
    {
        apply_weight_cos_principal_ray((float*)image, image.size(0), image.size(1), P);
    }
#endif

private:
    void run(float* images, int n, int n_u, int n_v, bool do_process, const double* Ps, ecc_ctx* ctx) const
    {
        ecc_preprocess_config cfg;
        ecc_preprocess_defaults(&cfg);
        cfg.process = do_process ? 1 : 0;
        cfg.normalize = intensity.normalize ? 1 : 0;
        cfg.bias = intensity.bias;
        cfg.scale = intensity.scale;
        cfg.apply_log = intensity.apply_log ? 1 : 0;
        cfg.gaussian_sigma = lowpass.gaussian_sigma;
        cfg.half_kernel_width = lowpass.half_kernel_width;
        cfg.flip_u = image_geometry.flip_u ? 1 : 0;
        cfg.flip_v = image_geometry.flip_v ? 1 : 0;
        std::vector<int32_t> blanks(4 * border.blanks.size());
        for (int k = 0; k < 4; ++k) {
            cfg.zero[k] = border.zero[k];
            cfg.feather[k] = border.feather[k];
        }
        for (size_t b = 0; b < border.blanks.size(); ++b)
            for (int k = 0; k < 4; ++k) blanks[4 * b + k] = border.blanks[b][k];
        cfg.n_blanks = (int32_t)border.blanks.size();
        cfg.blanks = blanks.empty() ? nullptr : blanks.data();
        detail::check(ecc_preprocess(ctx ? ctx : detail::default_context(), images, 0, images, n, n_u, n_v, &cfg, Ps));
    }
};

/// ref: EpipolarConsistency.h:36-46 (free functions).  estimateIsoCenter returns the point as (x, y, z, 1);
/// estimateAngularRange takes the two matrices whose baseline the reference's RP3Line argument is
/// (join_pluecker of their source positions, ref: EpipolarConsistencyDirect.cpp:88-92).
inline std::vector<double> estimateIsoCenter(const std::vector<ProjectionMatrix>& Ps)
{
    std::vector<double> flat(12 * Ps.size()), O(4, 0.0);
    for (size_t i = 0; i < Ps.size(); ++i)
        for (int k = 0; k < 12; ++k) flat[12 * i + k] = Ps[i].data()[k];
    ecc_host_iso_center(flat.data(), (int)Ps.size(), O.data());
    return O;
}
inline double estimateObjectRadius(const ProjectionMatrix& P, int n_u, int n_v) { return ecc_host_object_radius(P.data(), n_u, n_v); }
inline std::pair<double, double> estimateAngularRange(const ProjectionMatrix& P0, const ProjectionMatrix& P1, double object_radius_mm)
{
    std::pair<double, double> r(0.0, 0.0);
    ecc_host_angular_range(P0.data(), P1.data(), object_radius_mm, &r.first, &r.second);
    return r;
}
inline double estimateAngularStep(const ProjectionMatrix& P0, const ProjectionMatrix& P1, int n_u, int n_v)
{
    return ecc_host_angular_step(P0.data(), P1.data(), n_u, n_v);
}

/// ref: class Metric (interface)
class Metric {
protected:
    double object_radius_mm;
    double dkappa;
    std::vector<ProjectionMatrix> Ps;

public:
    Metric() : object_radius_mm(0), dkappa(0) {}
    virtual ~Metric() {}
    virtual Metric& setObjectRadius(double radius_mm = 0) { object_radius_mm = radius_mm; return *this; }
    virtual double getObjectRadius() const = 0;
    virtual Metric& setEpipolarPlaneStep(double dkappa_rad = 0) { dkappa = dkappa_rad; return *this; }
    virtual Metric& setProjectionMatrices(const std::vector<ProjectionMatrix>& _Ps) { Ps = _Ps; return *this; }
    const std::vector<ProjectionMatrix>& getProjectionMatrices() const { return Ps; }
    virtual int getNumberOfProjetions() = 0;
    virtual double evaluate(float* cost_image = 0x0) = 0;
};

/// ref: class MetricRadonIntermediate : public Metric
class MetricRadonIntermediate : public Metric {
    std::vector<RadonIntermediate*> dtrs;
    bool use_corr;
    int sampling;
    bool incremental;
    int record_reuse = -1;    // -1: the library's default
    int small_eval = -1;      // -1: the library's default (on)
    ecc_metric* m_h;          // single-device metric, or the group's rank-0 metric (borrowed) when m_gh is set
    ecc_group_metric* m_gh;   // sharded over a group of devices (ecc_group_*), else null
    ecc_ctx* m_ctx;
    ecc_group* m_group;

    void push_params()
    {
        if (m_gh) {
            detail::check(ecc_group_metric_set_params(m_gh, object_radius_mm, dkappa, use_corr ? 1 : 0));
            detail::check(ecc_group_metric_set_sampling(m_gh, sampling));
            detail::check(ecc_group_metric_set_incremental(m_gh, incremental ? 1 : 0));
            if (record_reuse >= 0 || small_eval >= 0) {  // per-metric switches: forwarded to every rank's metric
                ecc_group* g = m_group ? m_group : detail::default_group();
                const int ranks = g ? ecc_group_size(g) : 0;
                for (int r = 0; r < ranks; ++r) {
                    ecc_metric* rm = nullptr;
                    detail::check(ecc_group_metric_rank_metric(m_gh, r, &rm));
                    if (record_reuse >= 0) detail::check(ecc_metric_set_record_reuse(rm, record_reuse));
                    if (small_eval >= 0) detail::check(ecc_metric_set_small_eval(rm, small_eval));
                }
            }
        } else if (m_h) {
            detail::check(ecc_metric_set_params(m_h, object_radius_mm, dkappa, use_corr ? 1 : 0));
            detail::check(ecc_metric_set_sampling(m_h, sampling));
            detail::check(ecc_metric_set_incremental(m_h, incremental ? 1 : 0));
            if (record_reuse >= 0) detail::check(ecc_metric_set_record_reuse(m_h, record_reuse));
            if (small_eval >= 0) detail::check(ecc_metric_set_small_eval(m_h, small_eval));
        }
    }
    void push_projections()
    {
        if ((!m_h && !m_gh) || Ps.empty()) return;
        std::vector<double> flat(12 * Ps.size());
        for (size_t i = 0; i < Ps.size(); ++i)
            for (int k = 0; k < 12; ++k) flat[12 * i + k] = Ps[i].data()[k];
        if (m_gh) detail::check(ecc_group_metric_set_projections(m_gh, flat.data(), (int)Ps.size()));
        else detail::check(ecc_metric_set_projections(m_h, flat.data(), (int)Ps.size()));
    }
    /// the metric that serves the few-pair calls: with a group, rank 0's (the call also hands over pending matrices)
    ecc_metric* single()
    {
        if (m_gh) detail::check(ecc_group_metric_rank_metric(m_gh, 0, &m_h));
        return m_h;
    }
    void destroy()
    {
        if (m_gh) ecc_group_metric_destroy(m_gh);
        else ecc_metric_destroy(m_h);
        m_gh = nullptr;
        m_h = nullptr;
    }

public:
    explicit MetricRadonIntermediate(ecc_ctx* ctx = nullptr)
        : use_corr(false), sampling(ECC_SAMPLING_AUTO), incremental(false), m_h(nullptr), m_gh(nullptr), m_ctx(ctx), m_group(nullptr)
    {
    }
    /// ref: MetricRadonIntermediate(Ps, dtrs) (.h:31).  ctx: a context other than the process-wide default one.
    MetricRadonIntermediate(const std::vector<ProjectionMatrix>& _Ps, const std::vector<RadonIntermediate*>& _dtrs,
                            ecc_ctx* ctx = nullptr)
        : use_corr(false), sampling(ECC_SAMPLING_AUTO), incremental(false), m_h(nullptr), m_gh(nullptr), m_ctx(ctx), m_group(nullptr)
    {
        setProjectionMatrices(_Ps);
        setRadonIntermediates(_dtrs);
    }
    /// Same over an explicit group of devices (not in the reference): evaluate() is sharded over the group.
    MetricRadonIntermediate(const std::vector<ProjectionMatrix>& _Ps, const std::vector<RadonIntermediate*>& _dtrs,
                            ecc_group* group)
        : use_corr(false), sampling(ECC_SAMPLING_AUTO), incremental(false), m_h(nullptr), m_gh(nullptr), m_ctx(nullptr), m_group(group)
    {
        setProjectionMatrices(_Ps);
        setRadonIntermediates(_dtrs);
    }
    ~MetricRadonIntermediate() { destroy(); }

    MetricRadonIntermediate& setdKappa(float _dkappa) { dkappa = _dkappa; push_params(); return *this; }
    MetricRadonIntermediate& useCorrelation(bool corr = true) { use_corr = corr; push_params(); return *this; }
    /// Not in the reference: ECC_SAMPLING_* of the C ABI (default ECC_SAMPLING_AUTO).
    MetricRadonIntermediate& setSampling(int mode) { sampling = mode; push_params(); return *this; }
    /// Not in the reference: ecc_metric_set_incremental -- evaluate() re-evaluates only the pairs of views whose matrix
    /// changed since the last call (Gui/SingleImageMotion.h moves one view per call); bit-identical results.
    MetricRadonIntermediate& setIncremental(bool on = true) { incremental = on; push_params(); return *this; }
    /// Not in the reference: ecc_metric_set_record_reuse (library default: on) -- the per-pair geometry of pairs whose
    /// matrices did not change is kept between evaluate() calls; every pair is still sampled, bit-identical results.
    MetricRadonIntermediate& setRecordReuse(bool on = true, bool always = false) { record_reuse = on ? (always ? 2 : 1) : 0; push_params(); return *this; }
    /// Not in the reference: ecc_metric_set_small_eval (library default: on) -- evaluations of a few pairs (index lists,
    /// evaluate() of a handful of views: the FluoroTracking pattern) go out as ONE kernel launch; bit-identical results.
    MetricRadonIntermediate& setSmallEval(bool on = true) { small_eval = on ? 1 : 0; push_params(); return *this; }
    /// Not in the reference: ecc_metric_evaluate_poses -- independent all-pairs evaluations of several sets of projection
    /// matrices (a sweep as in Gui/Visualization.h:78-98 plotCostFunction, the probes of a finite-difference gradient),
    /// each value bit-identical to setProjectionMatrices(poses[k]) + evaluate().  Poses that differ from the current matrices
    /// (or from the first pose) in a few views are evaluated as ONE batched record / pair / sum launch each (ecc_poses.hip:
    /// two orders of magnitude above one call per pose).  The last pose stays current.
    std::vector<double> evaluatePoses(const std::vector<std::vector<Geometry::ProjectionMatrix> >& poses)
    {
        std::vector<double> means(poses.size(), 0.0);
        if (poses.empty()) return means;
        const size_t n = poses[0].size();
        std::vector<double> flat(12 * n * poses.size());
        for (size_t p = 0; p < poses.size(); ++p) {
            if (poses[p].size() != n) throw std::runtime_error("evaluatePoses: all poses need the same number of views");
            for (size_t i = 0; i < n; ++i)
                for (int k = 0; k < 12; ++k) flat[12 * (n * p + i) + k] = poses[p][i].data()[k];
        }
        if (m_gh) detail::check(ecc_group_metric_evaluate_poses(m_gh, (int)poses.size(), flat.data(), (int)n, means.data()));
        else detail::check(ecc_metric_evaluate_poses(m_h, (int)poses.size(), flat.data(), (int)n, means.data()));
        Ps = poses.back();
        return means;
    }

    /// Not in the reference: ecc_metric_evaluate_pose_deltas -- K poses that each replace a few views of the CURRENT matrices
    /// (moved_views[k]: strictly ascending view indices; moved_Ps[k]: their matrices), evaluated as one batched record / pair /
    /// sum launch each (a sweep of one view as in Gui/Visualization.h:59-112, the probes of a finite-difference gradient):
    /// every value bit-identical to replacing those views, setProjectionMatrices and evaluate(); the current matrices stay.
    /// With a device group the poses are expanded and dealt to the ranks (ecc_group_metric_evaluate_poses).
    std::vector<double> evaluatePoseDeltas(const std::vector<std::vector<int> >& moved_views,
                                           const std::vector<std::vector<Geometry::ProjectionMatrix> >& moved_Ps)
    {
        if (moved_views.size() != moved_Ps.size()) throw std::runtime_error("evaluatePoseDeltas: one matrix list per pose");
        std::vector<double> means(moved_views.size(), 0.0);
        if (moved_views.empty()) return means;
        if (m_gh) {
            std::vector<std::vector<Geometry::ProjectionMatrix> > poses(moved_views.size(), Ps);
            for (size_t k = 0; k < moved_views.size(); ++k) {
                if (moved_views[k].size() != moved_Ps[k].size()) throw std::runtime_error("evaluatePoseDeltas: one matrix per moved view");
                for (size_t q = 0; q < moved_views[k].size(); ++q) poses[k].at((size_t)moved_views[k][q]) = moved_Ps[k][q];
            }
            const std::vector<Geometry::ProjectionMatrix> keep = Ps;
            means = evaluatePoses(poses);
            setProjectionMatrices(keep);
            return means;
        }
        std::vector<int32_t> off(1, 0), views;
        std::vector<double> flat;
        for (size_t k = 0; k < moved_views.size(); ++k) {
            if (moved_views[k].size() != moved_Ps[k].size()) throw std::runtime_error("evaluatePoseDeltas: one matrix per moved view");
            for (size_t q = 0; q < moved_views[k].size(); ++q) {
                views.push_back(moved_views[k][q]);
                flat.insert(flat.end(), moved_Ps[k][q].data(), moved_Ps[k][q].data() + 12);
            }
            off.push_back((int32_t)views.size());
        }
        detail::check(ecc_metric_evaluate_pose_deltas(m_h, (int)moved_views.size(), off.data(), views.empty() ? nullptr : views.data(),
                                                      flat.empty() ? nullptr : flat.data(), means.data()));
        return means;
    }

    /// The metric borrows the dtrs: "DO NOT delete or change _dtrs during lifetime" (ref: .h:45).
    MetricRadonIntermediate& setRadonIntermediates(const std::vector<RadonIntermediate*>& _dtrs)
    {
        dtrs = _dtrs;
        destroy();
        std::vector<ecc_dtr*> hs(dtrs.size());
        for (size_t i = 0; i < dtrs.size(); ++i) hs[i] = dtrs[i]->handle();
        ecc_group* group = m_group ? m_group : (m_ctx ? nullptr : detail::default_group());
        if (group) detail::check(ecc_group_metric_create(group, (int)hs.size(), hs.data(), &m_gh));
        else detail::check(ecc_metric_create(m_ctx ? m_ctx : detail::default_context(), (int)hs.size(), hs.data(), &m_h));
        push_params();
        push_projections();
        return *this;
    }
    const std::vector<RadonIntermediate*>& getRadonIntermediates() const { return dtrs; }

    virtual Metric& setObjectRadius(double radius_mm = 0) { Metric::setObjectRadius(radius_mm); push_params(); return *this; }
    virtual Metric& setEpipolarPlaneStep(double dkappa_rad = 0) { Metric::setEpipolarPlaneStep(dkappa_rad); push_params(); return *this; }
    virtual double getObjectRadius() const
    {
        double r = 0;
        if (m_gh) detail::check(ecc_group_metric_get_object_radius(m_gh, &r));
        else if (m_h) detail::check(ecc_metric_get_object_radius(m_h, &r));
        return r;
    }

    virtual Metric& setProjectionMatrices(const std::vector<ProjectionMatrix>& _Ps)
    {
        Metric::setProjectionMatrices(_Ps);
        push_projections();
        return *this;
    }

    virtual int getNumberOfProjetions() { return (int)Ps.size(); }

    /// Out is n*n (index i + j*n, i<j written, the rest preserved) and the mean is returned.
    virtual double evaluate(float* out = 0x0)
    {
        double mean = 0;
        if (m_gh) detail::check(ecc_group_metric_evaluate_all(m_gh, out, &mean));
        else detail::check(ecc_metric_evaluate_all(m_h, out, &mean));
        return mean;
    }

    /// ref: evaluate(const std::set<int>& views, float* out): all pairs inside the subset.
    double evaluate(const std::set<int>& views, float* _out = 0x0)
    {
        std::vector<int32_t> idx;
        for (std::set<int>::const_iterator i = views.begin(); i != views.end(); ++i)
            for (std::set<int>::const_iterator j = i; j != views.end(); ++j) {
                if (i == j) continue;
                const int32_t t[4] = {*i, *j, *i, *j};
                idx.insert(idx.end(), t, t + 4);
            }
        return evaluate_indices(idx.data(), (int)(idx.size() / 4), _out);
    }

#ifdef ECC_ADAPTER_HAVE_EIGEN
    /// ref: evaluate(const std::vector<Eigen::Vector4i>& indices, float* out): (P0, P1, dtr0, dtr1) tuples.
    double evaluate(const std::vector<Eigen::Vector4i>& _indices, float* _out)
    {
        std::vector<int32_t> idx(4 * _indices.size());
        for (size_t i = 0; i < _indices.size(); ++i)
            for (int k = 0; k < 4; ++k) idx[4 * i + k] = _indices[i][k];
        return evaluate_indices(idx.data(), (int)_indices.size(), _out);
    }
#endif
    /// Same with a flat int32 array of 4-tuples (what the Eigen overload forwards to).
    double evaluate_indices(const int32_t* idx4, int n_pairs, float* _out)
    {
        double mean = 0;
        detail::check(ecc_metric_evaluate_pairs(single(), idx4, n_pairs, _out, &mean));
        return mean;
    }

    /// ref: evaluateForImagePair(i, j, redundant_samples0, redundant_samples1, kappas) -- visualisation.
    virtual double evaluateForImagePair(int i, int j, std::vector<float>* redundant_samples0 = 0x0,
                                        std::vector<float>* redundant_samples1 = 0x0, std::vector<float>* kappas = 0x0)
    {
        return evaluateForImagePair(i, j, redundant_samples0, redundant_samples1, kappas, 0x0, 0x0);
    }

    /// ref: ... and the sample locations in the two Radon transforms (angle, distance in [0,1]).
    double evaluateForImagePair(int i, int j, std::vector<float>* redundant_samples0,
                                std::vector<float>* redundant_samples1, std::vector<float>* kappas,
                                std::vector<std::pair<float, float> >* radon_samples0,
                                std::vector<std::pair<float, float> >* radon_samples1)
    {
        int cap = 0, n = 0;
        detail::check(ecc_metric_pair_samples_bound(single(), &cap));
        std::vector<float> s0(cap), s1(cap), kp(cap), r0(2 * (size_t)cap), r1(2 * (size_t)cap);
        double ecc = 0;
        detail::check(ecc_metric_evaluate_for_image_pair(m_h, i, j, cap, &n, s0.data(), s1.data(), kp.data(), r0.data(),
                                                         r1.data(), 0x0, &ecc));
        if (redundant_samples0) redundant_samples0->insert(redundant_samples0->end(), s0.begin(), s0.begin() + n);
        if (redundant_samples1) redundant_samples1->insert(redundant_samples1->end(), s1.begin(), s1.begin() + n);
        if (kappas) kappas->insert(kappas->end(), kp.begin(), kp.begin() + n);
        for (int k = 0; k < n; ++k) {
            if (radon_samples0) radon_samples0->push_back(std::make_pair(r0[2 * k], r0[2 * k + 1]));
            if (radon_samples1) radon_samples1->push_back(std::make_pair(r1[2 * k], r1[2 * k + 1]));
        }
        return ecc;
    }

private:
    MetricRadonIntermediate(const MetricRadonIntermediate&);
};

/// ref: class MetricDirect : public Metric (EpipolarConsistencyDirect.h:28-60).  The reference takes
/// UtilsCuda::BindlessTexture2D<float>* per view; here the projection stack is one float array, on the host
/// (copied to the device) or already on the device (borrowed).
class MetricDirect : public Metric {
    ecc_direct* m_h;
    int n_images;
    bool use_fbcc;

    void push_params() { detail::check(ecc_direct_set_params(m_h, object_radius_mm, dkappa, use_fbcc ? 1 : 0)); }

public:
    MetricDirect(const std::vector<ProjectionMatrix>& _Ps, const float* images, int n, int n_u, int n_v,
                 bool images_on_device = false, ecc_ctx* ctx = nullptr)
        : m_h(nullptr), n_images(n), use_fbcc(false)
    {
        detail::check(ecc_direct_create(ctx ? ctx : detail::default_context(), n, images, images_on_device ? 1 : 0, n_u,
                                        n_v, &m_h));
        setProjectionMatrices(_Ps);
    }
    ~MetricDirect() { ecc_direct_destroy(m_h); }

    virtual Metric& setProjectionMatrices(const std::vector<ProjectionMatrix>& _Ps)
    {
        Metric::setProjectionMatrices(_Ps);
        if (Ps.empty()) return *this;
        std::vector<double> flat(12 * Ps.size());
        for (size_t i = 0; i < Ps.size(); ++i)
            for (int k = 0; k < 12; ++k) flat[12 * i + k] = Ps[i].data()[k];
        detail::check(ecc_direct_set_projections(m_h, flat.data(), (int)Ps.size()));
        return *this;
    }
    virtual Metric& setObjectRadius(double radius_mm = 0) { Metric::setObjectRadius(radius_mm); push_params(); return *this; }
    virtual Metric& setEpipolarPlaneStep(double dkappa_rad = 0) { Metric::setEpipolarPlaneStep(dkappa_rad); push_params(); return *this; }
    virtual double getObjectRadius() const
    {
        double r = 0;
        detail::check(ecc_direct_get_object_radius(m_h, &r));
        return r;
    }
    virtual int getNumberOfProjetions() { return n_images; }

    /// Evaluates the metric (SUM over pairs, as the reference does) and optionally returns the n*n cost image.
    virtual double evaluate(float* out = 0x0)
    {
        double sum = 0;
        detail::check(ecc_direct_evaluate(m_h, out, &sum));
        return sum;
    }

    virtual double evaluateForImagePair(int i, int j, std::vector<float>* redundant_samples0 = 0x0,
                                        std::vector<float>* redundant_samples1 = 0x0, std::vector<float>* kappas = 0x0)
    {
        int cap = 0, n = 0;
        detail::check(ecc_direct_lines_bound(m_h, &cap));
        std::vector<float> s0(cap), s1(cap), kp(cap);
        double metric = 0;
        detail::check(ecc_direct_evaluate_for_image_pair(m_h, i, j, cap, &n, s0.data(), s1.data(), kp.data(), 0x0, &metric));
        if (redundant_samples0) redundant_samples0->assign(s0.begin(), s0.begin() + n);
        if (redundant_samples1) redundant_samples1->assign(s1.begin(), s1.begin() + n);
        if (kappas) kappas->assign(kp.begin(), kp.begin() + n);
        return metric;
    }

    /// ref: setFanBeamConsistency: use the rectified fan-beam weighting instead of the derivative.
    MetricDirect& setFanBeamConsistency(bool fbcc = true) { use_fbcc = fbcc; push_params(); return *this; }

private:
    MetricDirect(const MetricDirect&);
};

}  // namespace EpipolarConsistency

#endif
