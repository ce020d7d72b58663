// ref_shim.cpp -- extern "C" doorway onto the REFERENCE's own host-compilable headers.
// TEST INFRASTRUCTURE ONLY (see ecc_oracle.c header).  Built by oracle/Makefile into
// oracle/_ref/libecc_ref.so, compiling the headers where they lie under /root/reference
// (nothing is copied into this repository).  Used by tests/ to pin the oracle's E1/E2 geometry.
//
// <stdlib.h>/<math.h> are included first so that the unqualified abs() inside
// culaut/xgeinv.hxx:51-52 resolves to the floating-point overload, as it does under nvcc/MSVC
// (with <cmath> alone g++ would pick int abs(int) and truncate the column scale).
#include <stdlib.h>
#include <math.h>
#include <cmath>
#include <LibUtilsCuda/culaut/xprojectionmatrix.hxx>
#include <LibEpipolarConsistency/EpipolarConsistencyCommon.hxx>
#include <NRRD/nrrd_image.hxx>
#include <NRRD/nrrd_lowpass.hxx>

#define REF_API extern "C" __attribute__((visibility("default")))

REF_API void ref_pinvT(const double* P, float* PinvT)
{
	culaut::projection_matrix_pseudoinverse_transpose<double, float>(P, PinvT);
}

REF_API void ref_source_position(const double* P, float* C)
{
	culaut::projection_matrix_source_position<double, float>(P, C);
}

REF_API void ref_get_ij(int ij, int n, int* i, int* j)
{
	short si, sj;
	get_ij(ij, (short)n, si, sj);
	*i = si;
	*j = sj;
}

REF_API void ref_computeK01(float n_x2, float n_y2, float* C0, float* C1, float* P0invT, float* P1invT,
	float object_radius_mm, float num_samples, float dkappa, float* K0, float* K1)
{
	computeK01(n_x2, n_y2, C0, C1, P0invT, P1invT, object_radius_mm, num_samples, dkappa, K0, K1);
}

REF_API int ref_line_to_sample_dtr(float* line, float range_t)
{
	return lineToSampleDtr(line, range_t) ? 1 : 0;
}

REF_API float ref_weighting(float x)
{
	return weighting<float>(x);
}

// ---- the pre-processing low-pass and the host image interpolation (header-only NRRD library) ----
// ref: HeaderOnly/NRRD/nrrd_lowpass.hxx:19-34 (gaussianKernel)
REF_API void ref_gaussian_kernel(double sigma, int k, double* kernel)
{
	std::vector<double> g = NRRD::gaussianKernel(sigma, k);
	for (size_t i = 0; i < g.size(); i++) kernel[i] = g[i];
}

// ref: HeaderOnly/NRRD/nrrd_lowpass.hxx:183-190 (lowpass2D -> convolve2D :45-79), in place on the caller's image
REF_API void ref_lowpass2D(float* img, int w, int h, double sigma, int k)
{
	NRRD::ImageView<float> view(w, h, 1, img);
	NRRD::lowpass2D(view, sigma, k);
}

// ref: HeaderOnly/NRRD/nrrd_image_view.hxx:189-210 (ImageView::operator()(double, double, double)), 2D
REF_API double ref_image_view_at(float* img, int w, int h, double x, double y)
{
	NRRD::ImageView<float> view(w, h, 1, img);
	return view(x, y);
}
