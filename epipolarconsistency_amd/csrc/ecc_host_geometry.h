// ecc_host_geometry.h -- per-view host pre-compute (E1) and default object radius (E5), float64.
//
// Implements what MetricRadonIntermediate::setProjectionMatrices computes per view
// (ref: LibEpipolarConsistency/EpipolarConsistencyRadonIntermediate.cpp:134-163):
//   (P^+)^T  = ((P P^T)^-1 P)  as 3x4 column-major float   (ref: culaut/xprojectionmatrix.hxx:20-52)
//   C        = null vector of P scaled to w = 1, float       (ref: culaut/xprojectionmatrix.hxx:93-105)
// and Metric::getObjectRadius' estimate (ref: EpipolarConsistency.cpp:35-47,76-84).
//
// P P^T of a C-arm projection matrix has a condition number of ~1e11, so the float32 results
// depend on HOW the inverse is formed.  To be a drop-in, householder_qr / back_substitute below
// follow the reference's culaut routines (xsqqr / xutsolve / xgeinv, xgeinv.hxx:40-168) step for
// step -- the same Householder QR with the same operation order in binary64, including its
// never-reset running column scale: any other order gives different float32 bits at this
// condition number.  Bit equality with the reference headers (compiled where they lie) is
// checked in tests/test_oracle_pins.py; everything around these ~70 lines is this project's.
#ifndef ECC_HOST_GEOMETRY_H
#define ECC_HOST_GEOMETRY_H

#include <cmath>

// The same source is compiled for the host (C ABI helpers, tests) and for the device (e1_kernel in
// geometry_kernel.hip): binary64 add/mul/div/sqrt are correctly rounded on both and contraction is
// off, so both give bit-identical results.
#if defined(__HIPCC__)
#define ECC_HD __host__ __device__
#else
#define ECC_HD
#endif

namespace ecc_host {

// In-place Householder QR of the column-major N x N matrix M (M becomes R) with explicit Q.
// Arithmetic contract (must not be re-ordered):
//   * column scale = running maximum of |M(i,k)|, i >= k, over columns 0..k (never reset);
//   * reflector k: v = M(k:N,k)/scale, sigma = sign(v_k)*||v||, v_k += sigma, c_k = sigma*v_k,
//     diag_k = -scale*sigma; trailing columns j: M(k:N,j) -= (v.M(k:N,j) / c_k) * v;
//   * last diagonal entry is negated; Q accumulates reflectors 0..N-2 applied to the identity.
template <int N>
ECC_HD inline void householder_qr(double* M, double* Q)
{
    double diag[N], c[N];
    double scale = 0.0;
    for (int k = 0; k < N; ++k) {
        double* vk = M + N * k;
        for (int i = k; i < N; ++i) {
            const double a = fabs(vk[i]);
            if (scale < a) scale = a;
        }
        if (scale == 0.0) {
            c[k] = diag[k] = 0.0;
            continue;
        }
        for (int i = k; i < N; ++i) vk[i] /= scale;
        double nrm2 = 0.0;
        for (int i = k; i < N; ++i) nrm2 += vk[i] * vk[i];
        const double sigma = vk[k] > 0.0 ? sqrt(nrm2) : -sqrt(nrm2);
        vk[k] += sigma;
        c[k] = sigma * vk[k];
        diag[k] = -scale * sigma;
        for (int j = k + 1; j < N; ++j) {
            double* vj = M + N * j;
            double dot = 0.0;
            for (int i = k; i < N; ++i) dot += vk[i] * vj[i];
            const double tau = dot / c[k];
            for (int i = k; i < N; ++i) vj[i] -= tau * vk[i];
        }
    }
    diag[N - 1] *= -1;
    for (int i = 0; i < N; ++i) {
        for (int j = 0; j < N; ++j) Q[i + N * j] = 0.0;
        Q[i + N * i] = 1.0;
    }
    for (int k = 0; k < N - 1; ++k) {
        if (c[k] == 0.0) continue;
        const double* vk = M + N * k;
        for (int j = 0; j < N; ++j) {
            double dot = 0.0;
            for (int i = k; i < N; ++i) dot += vk[i] * Q[j + N * i];
            dot /= c[k];
            for (int i = k; i < N; ++i) Q[j + N * i] -= dot * vk[i];
        }
    }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            if (i == j) M[i + N * i] = diag[i];
            else if (i > j) M[i + N * j] = 0.0;
        }
}

// x = R^-1 b for upper-triangular column-major R (back substitution, last row first).
template <int N>
ECC_HD inline void back_substitute(const double* R, const double* b, double* x)
{
    x[N - 1] = b[N - 1] / R[(N - 1) * N + (N - 1)];
    for (int i = N - 2; i >= 0; --i) {
        x[i] = b[i];
        for (int j = i + 1; j < N; ++j) x[i] -= R[j * N + i] * x[j];
        x[i] = x[i] / R[i * N + i];
    }
}

// (P^+)^T, 3x4 column-major, double in -> float out.  P: 3x4 column-major.
ECC_HD inline void pinv_transpose(const double* P, float* out12)
{
    double G[9];  // Gram matrix P P^T (symmetric)
    G[0] = P[0] * P[0] + P[3] * P[3] + P[6] * P[6] + P[9] * P[9];
    G[1] = P[0] * P[1] + P[3] * P[4] + P[6] * P[7] + P[9] * P[10];
    G[2] = P[0] * P[2] + P[3] * P[5] + P[6] * P[8] + P[9] * P[11];
    G[4] = P[1] * P[1] + P[4] * P[4] + P[7] * P[7] + P[10] * P[10];
    G[5] = P[1] * P[2] + P[4] * P[5] + P[7] * P[8] + P[10] * P[11];
    G[8] = P[2] * P[2] + P[5] * P[5] + P[8] * P[8] + P[11] * P[11];
    G[3] = G[1];
    G[6] = G[2];
    G[7] = G[5];
    double Q[9], Ginv[9];
    householder_qr<3>(G, Q);  // G := R
    for (int col = 0; col < 3; ++col) {
        double e[3] = {0.0, 0.0, 0.0}, qtb[3];
        e[col] = 1.0;
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int i = 0; i < 3; ++i) s += e[i] * Q[i + 3 * j];
            qtb[j] = s;
        }
        back_substitute<3>(G, qtb, Ginv + 3 * col);
    }
    for (int c4 = 0; c4 < 4; ++c4) {
        const double* p = P + 3 * c4;
        for (int r = 0; r < 3; ++r)
            out12[3 * c4 + r] = (float)(p[0] * Ginv[3 * r + 0] + p[1] * Ginv[3 * r + 1] + p[2] * Ginv[3 * r + 2]);
    }
}

// Source position: last column of Q in the QR of the 4x4 matrix (P^T | 0), scaled to w = 1.
ECC_HD inline void source_position(const double* P, float* out4)
{
    double A[16], Q[16];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) A[c + 4 * r] = P[r + 3 * c];
    for (int c = 0; c < 4; ++c) A[c + 12] = 0.0;
    householder_qr<4>(A, Q);
    for (int i = 0; i < 4; ++i) out4[i] = (float)(Q[i + 12] / Q[15]);
}

ECC_HD inline void cross3(const double* a, const double* b, double* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
ECC_HD inline double det3(const double* a, const double* b, const double* c)
{
    return a[0] * (b[1] * c[2] - b[2] * c[1]) - b[0] * (a[1] * c[2] - a[2] * c[1]) +
           c[0] * (a[1] * b[2] - a[2] * b[1]);
}

// ref: EpipolarConsistency.cpp:35-47 with getCameraFocalLengthPx (ProjectionMatrix.cpp:104-112)
// and getCameraCenter (:70-76; the reference uses an SVD null space, any float64 null space
// agrees to ~1e-13 -- here signed 3x3 minors).
ECC_HD inline double object_radius(const double* P, int n_u, int n_v)
{
    const double m1[3] = {P[0], P[3], P[6]}, m2[3] = {P[1], P[4], P[7]}, m3[3] = {P[2], P[5], P[8]};
    double U[3], V[3], t[3];
    cross3(m3, m2, U);
    double n = sqrt(U[0] * U[0] + U[1] * U[1] + U[2] * U[2]);
    U[0] /= n; U[1] /= n; U[2] /= n;
    cross3(m3, m1, V);
    n = sqrt(V[0] * V[0] + V[1] * V[1] + V[2] * V[2]);
    V[0] /= n; V[1] /= n; V[2] /= n;
    cross3(V, m3, t);
    const double fu = m1[0] * t[0] + m1[1] * t[1] + m1[2] * t[2];
    cross3(U, m3, t);
    const double fv = m2[0] * t[0] + m2[1] * t[1] + m2[2] * t[2];
    const double a = fabs(atan(0.5 * n_u / fu)), b = fabs(atan(0.5 * n_v / fv));
    const double fov = a > b ? a : b;
    double C[4] = {det3(P + 3, P + 6, P + 9), -det3(P, P + 6, P + 9), det3(P, P + 3, P + 9), -det3(P, P + 3, P + 6)};
    if (C[3] < -1e-12 || C[3] > 1e-12) {
        C[0] /= C[3]; C[1] /= C[3]; C[2] /= C[3];
    }
    return sin(fov) * sqrt(C[0] * C[0] + C[1] * C[1] + C[2] * C[2]);
}

// Intrinsics the cosine weighting needs: K(0,0), K(0,2), K(1,2) of P = K [R|t], K upper triangular with
// positive diagonal and K(2,2) = 1 (ref: LibProjectiveGeometry/ProjectionMatrix.cpp:25-67, getCameraIntrinsics;
// used at Gui/PreProccess.cpp:152-155).  The reference factorises with Eigen's Householder QR; this is the same
// RQ factorisation by Gram-Schmidt on the rows of M = P(:,0:3) in binary64 (agrees to ~1e-13).
ECC_HD inline void intrinsics(const double* P, float* sdd_px, float* ppu, float* ppv)
{
    const double m1[3] = {P[0], P[3], P[6]}, m2[3] = {P[1], P[4], P[7]}, m3[3] = {P[2], P[5], P[8]};
    const double K22 = sqrt(m3[0] * m3[0] + m3[1] * m3[1] + m3[2] * m3[2]);
    const double r3[3] = {m3[0] / K22, m3[1] / K22, m3[2] / K22};
    const double K12 = m2[0] * r3[0] + m2[1] * r3[1] + m2[2] * r3[2];
    const double v[3] = {m2[0] - K12 * r3[0], m2[1] - K12 * r3[1], m2[2] - K12 * r3[2]};
    const double K11 = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const double r2[3] = {v[0] / K11, v[1] / K11, v[2] / K11};
    const double K02 = m1[0] * r3[0] + m1[1] * r3[1] + m1[2] * r3[2];
    const double K01 = m1[0] * r2[0] + m1[1] * r2[1] + m1[2] * r2[2];
    const double w[3] = {m1[0] - K02 * r3[0] - K01 * r2[0], m1[1] - K02 * r3[1] - K01 * r2[1],
                         m1[2] - K02 * r3[2] - K01 * r2[2]};
    const double K00 = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    *sdd_px = (float)(K00 / K22);
    *ppu = (float)(K02 / K22);
    *ppv = (float)(K12 / K22);
}

// ---- MetricDirect geometry (float64 throughout, like the reference's host code) -------------------------
// (P^+)^T E for planes E: the reference forms P^+ with Eigen's JacobiSVD (ref:
// LibProjectiveGeometry/SingularValueDecomposition.cpp:10-25).  Same quantity by modified Gram-Schmidt on the
// rows of P: P = L Q (L lower triangular, Q with orthonormal rows), (P^+)^T = L^-T Q; works on P itself
// (condition ~3e5), not on P P^T (~1e11).
struct RowQR {
    double Q[3][4];
    double L[3][3];
};

ECC_HD inline void row_qr(const double* P, RowQR* f)
{
    for (int i = 0; i < 3; ++i) {
        double v[4];
        for (int k = 0; k < 4; ++k) v[k] = P[i + 3 * k];
        for (int j = 0; j < 3; ++j) f->L[i][j] = 0.0;
        for (int j = 0; j < i; ++j) {
            double dot = 0;
            for (int k = 0; k < 4; ++k) dot += v[k] * f->Q[j][k];
            f->L[i][j] = dot;
            for (int k = 0; k < 4; ++k) v[k] -= dot * f->Q[j][k];
        }
        const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
        f->L[i][i] = n;
        for (int k = 0; k < 4; ++k) f->Q[i][k] = v[k] / n;
    }
}

ECC_HD inline void plane_to_line(const RowQR& f, const double* E, double* l)
{
    double y[3];
    for (int i = 0; i < 3; ++i) {
        y[i] = 0;
        for (int k = 0; k < 4; ++k) y[i] += f.Q[i][k] * E[k];
    }
    l[2] = y[2] / f.L[2][2];
    l[1] = (y[1] - f.L[2][1] * l[2]) / f.L[1][1];
    l[0] = (y[0] - f.L[1][0] * l[1] - f.L[2][0] * l[2]) / f.L[0][0];
}

// ref: LibProjectiveGeometry/ProjectionMatrix.cpp:70-76 (getCameraCenter; any float64 null space), w = 1.
ECC_HD inline void camera_center(const double* P, double* C)
{
    C[0] = det3(P + 3, P + 6, P + 9);
    C[1] = -det3(P, P + 6, P + 9);
    C[2] = det3(P, P + 3, P + 9);
    C[3] = -det3(P, P + 3, P + 6);
    if (C[3] < -1e-12 || C[3] > 1e-12) {
        C[0] /= C[3]; C[1] /= C[3]; C[2] /= C[3]; C[3] = 1.0;
    }
}

// ref: LibProjectiveGeometry/ProjectiveGeometry.hxx:188-200, 216-224 (Pluecker joins)
ECC_HD inline void join_points(const double* A, const double* B, double* L)
{
    L[0] = A[0] * B[1] - A[1] * B[0];
    L[1] = A[0] * B[2] - A[2] * B[0];
    L[2] = A[0] * B[3] - A[3] * B[0];
    L[3] = A[1] * B[2] - A[2] * B[1];
    L[4] = A[1] * B[3] - A[3] * B[1];
    L[5] = A[2] * B[3] - A[3] * B[2];
}
ECC_HD inline void join_line_point(const double* L, const double* X, double* E)
{
    E[0] = +X[1] * L[5] - X[2] * L[4] + X[3] * L[3];
    E[1] = -X[0] * L[5] + X[2] * L[2] - X[3] * L[1];
    E[2] = +X[0] * L[4] - X[1] * L[2] + X[3] * L[0];
    E[3] = -X[0] * L[3] + X[1] * L[1] - X[2] * L[0];
}

// ---- rectified fan-beam consistency (FBCC), ref: RectifiedFBCC.h, EpipolarConsistencyDirect.cpp:133-196 ----
// ref: ProjectiveGeometry.hxx:202-214 (meet of two planes -> line), :226-235 (meet of a line and a plane)
ECC_HD inline void meet_planes(const double* A, const double* B, double* L)
{
    L[0] = A[2] * B[3] - A[3] * B[2];
    L[1] = A[3] * B[1] - A[1] * B[3];
    L[2] = A[1] * B[2] - A[2] * B[1];
    L[3] = A[0] * B[3] - A[3] * B[0];
    L[4] = A[2] * B[0] - A[0] * B[2];
    L[5] = A[0] * B[1] - A[1] * B[0];
}
ECC_HD inline void meet_line_plane(const double* L, const double* P, double* X)
{
    X[0] = -P[1] * L[0] - P[2] * L[1] - P[3] * L[2];
    X[1] = +P[0] * L[0] - P[2] * L[3] - P[3] * L[4];
    X[2] = +P[0] * L[1] + P[1] * L[3] - P[3] * L[5];
    X[3] = +P[0] * L[2] + P[1] * L[4] + P[2] * L[5];
}
// ref: ProjectiveGeometry.hxx:38-54, 75-91 (dehomogenize)
ECC_HD inline void dehom3(double* X)
{
    if (X[3] > 1e-12 || X[3] < -1e-12) { X[0] /= X[3]; X[1] /= X[3]; X[2] /= X[3]; X[3] = 1; }
    else { X[3] = 0; const double n = sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]); X[0] /= n; X[1] /= n; X[2] /= n; }
}
ECC_HD inline void dehom2(double* x)
{
    if (x[2] > 1e-11 || x[2] < -1e-11) { x[0] /= x[2]; x[1] /= x[2]; x[2] = 1; }
    else { x[2] = 0; const double n = sqrt(x[0] * x[0] + x[1] * x[1]); x[0] /= n; x[1] /= n; }
}

// H = P_E * centralProjectionToPlane(C, E) * P^+ (3 x 3, row-major), P^+ = Q^T L^-1 from the row-QR
// (ref: ...Direct.cpp:143-151, ProjectiveGeometry.hxx:333-342).
ECC_HD inline void fbcc_homography(const RowQR& f, const double* C, const double* U, const double* V, const double* E,
                                   double* H)
{
    double Linv[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, Pinv[4][3], CP[4][4], T[3][4], PE[3][4];
    for (int i = 0; i < 3; ++i) {
        Linv[i][i] = 1.0 / f.L[i][i];
        for (int j = 0; j < i; ++j) {
            double sum = 0;
            for (int k = j; k < i; ++k) sum += f.L[i][k] * Linv[k][j];
            Linv[i][j] = -sum / f.L[i][i];
        }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j) {
            double sum = 0;
            for (int k = 0; k < 3; ++k) sum += f.Q[k][i] * Linv[k][j];
            Pinv[i][j] = sum;
        }
    CP[0][0] = +C[1] * E[1] + C[2] * E[2] + C[3] * E[3]; CP[0][1] = -C[0] * E[1]; CP[0][2] = -C[0] * E[2]; CP[0][3] = -C[0] * E[3];
    CP[1][0] = -C[1] * E[0]; CP[1][1] = +C[0] * E[0] + C[2] * E[2] + C[3] * E[3]; CP[1][2] = -C[1] * E[2]; CP[1][3] = -C[1] * E[3];
    CP[2][0] = -C[2] * E[0]; CP[2][1] = -C[2] * E[1]; CP[2][2] = +C[0] * E[0] + C[3] * E[3] + C[1] * E[1]; CP[2][3] = -C[2] * E[3];
    CP[3][0] = -C[3] * E[0]; CP[3][1] = -C[3] * E[1]; CP[3][2] = -C[3] * E[2]; CP[3][3] = +C[0] * E[0] + C[1] * E[1] + C[2] * E[2];
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 4; ++k) PE[i][k] = 0;
    for (int k = 0; k < 3; ++k) { PE[0][k] = U[k]; PE[1][k] = V[k]; }
    PE[2][3] = 1.0;  // pixel_spacing
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            double sum = 0;
            for (int k = 0; k < 4; ++k) sum += PE[i][k] * CP[k][j];
            T[i][j] = sum;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double sum = 0;
            for (int k = 0; k < 4; ++k) sum += T[i][k] * Pinv[k][j];
            H[3 * i + j] = sum;
        }
}

// ref: RectifiedFBCC.h:18-90 (LinePerspectivity + FBCC_weighting_info; float members, float arithmetic)
struct FbccInfo {
    float a, b, c, d, t_prime_ak, d_l_kappa_C_sq;
    ECC_HD float transform(float t) const { return (a * t + b) / (c * t + d); }
    ECC_HD float derivative(float t) const { return (a * d - b * c) / (c * c * t * t + 2 * c * d * t + d * d); }
};

// ref: ...Direct.cpp:153-191: per-line weighting info from the (float) epipolar line lf of the view (P, C, H).
ECC_HD inline void fbcc_line_info(const double* P, const double* C, const double* H, const double* dvec, const double* E,
                                  const float* lf, FbccInfo* out)
{
    const double l[3] = {lf[0], lf[1], lf[2]};
    double Ek[4], EB[4], M[6], Ak[4], ak[3], dist = 0;
    for (int k = 0; k < 4; ++k) Ek[k] = P[0 + 3 * k] * l[0] + P[1 + 3 * k] * l[1] + P[2 + 3 * k] * l[2];  // P^T l
    EB[0] = dvec[0]; EB[1] = dvec[1]; EB[2] = dvec[2];
    EB[3] = -(dvec[0] * C[0] + dvec[1] * C[1] + dvec[2] * C[2]);
    meet_planes(EB, Ek, M);
    meet_line_plane(M, E, Ak);
    dehom3(Ak);
    for (int k = 0; k < 4; ++k) {
        const double diff = Ak[k] - C[k];
        dist += diff * diff;
    }
    const float d_px = (float)(sqrt(dist) / 1.0);
    out->d_l_kappa_C_sq = d_px * d_px;
    for (int k = 0; k < 3; ++k) ak[k] = P[k + 0] * Ak[0] + P[k + 3] * Ak[1] + P[k + 6] * Ak[2] + P[k + 9] * Ak[3];
    dehom2(ak);
    const double a = H[0] * l[1] - H[3] * l[0];
    const double b = H[2] - H[0] * l[0] * l[2];
    const double c = H[6] * l[1] - H[7] * l[0];
    const double d = H[8] - H[6] * l[0] * l[2] - H[7] * l[1] * l[2];
    out->a = (float)a; out->b = (float)b; out->c = (float)c; out->d = (float)d;
    if (out->a * out->d - out->b * out->c < 0) { out->a *= -1; out->b *= -1; }
    const double t_ak = l[1] * ak[0] / ak[2] - l[0] * ak[1] / ak[2];
    out->t_prime_ak = out->transform((float)t_ak);
}

}  // namespace ecc_host
#endif
