// ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under ipc_amd/ may include, link or call this.
//
// Small dense helpers for the CPU restatement of the IPC Newton hot path.
// Matrices are column-major double arrays, M(i,j) = m[i + n*j], like Eigen's default
// storage in the reference ("Eigen default storage is column-major").
#pragma once
#include <cmath>
#include <cstring>
#include <algorithm>

namespace orc {

struct M3 {
    double m[9];
    double& operator()(int i, int j) { return m[i + 3 * j]; }
    double operator()(int i, int j) const { return m[i + 3 * j]; }
};

inline M3 mul(const M3& A, const M3& B)
{
    M3 C;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += A(i, k) * B(k, j);
            C(i, j) = s;
        }
    return C;
}
inline double det(const M3& A)
{
    return A(0, 0) * (A(1, 1) * A(2, 2) - A(1, 2) * A(2, 1))
        - A(0, 1) * (A(1, 0) * A(2, 2) - A(1, 2) * A(2, 0))
        + A(0, 2) * (A(1, 0) * A(2, 1) - A(1, 1) * A(2, 0));
}
// cofactor matrix, entry for entry as IglUtils::computeCofactorMtr (IglUtils.hpp:436-464)
inline M3 cofactor(const M3& F)
{
    M3 A;
    A(0, 0) = F(1, 1) * F(2, 2) - F(1, 2) * F(2, 1);
    A(0, 1) = F(1, 2) * F(2, 0) - F(1, 0) * F(2, 2);
    A(0, 2) = F(1, 0) * F(2, 1) - F(1, 1) * F(2, 0);
    A(1, 0) = F(0, 2) * F(2, 1) - F(0, 1) * F(2, 2);
    A(1, 1) = F(0, 0) * F(2, 2) - F(0, 2) * F(2, 0);
    A(1, 2) = F(0, 1) * F(2, 0) - F(0, 0) * F(2, 1);
    A(2, 0) = F(0, 1) * F(1, 2) - F(0, 2) * F(1, 1);
    A(2, 1) = F(0, 2) * F(1, 0) - F(0, 0) * F(1, 2);
    A(2, 2) = F(0, 0) * F(1, 1) - F(0, 1) * F(1, 0);
    return A;
}
inline M3 inverse(const M3& A)
{
    M3 C = cofactor(A);
    double d = det(A);
    M3 R;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R(i, j) = C(j, i) / d; // inverse = cof^T / det
    return R;
}

// 3x3 SVD  F = U diag(s) V^T with the output convention of the reference's
// JIXIE::singularValueDecomposition (ImplicitQRSVD.h:681-850, AutoFlipSVD.hpp:39-44):
// U, V proper rotations, |s0| >= |s1| >= |s2|, only s2 may be negative.
// The iteration is a one-sided (Hestenes) Jacobi instead of the implicit-shift QR
// of ImplicitQRSVD.h:687-850: every quantity the hot path derives from the SVD
// (psi, P, dP/dF after projection) is a function of F alone, so the sweep scheme
// only moves results at round-off level.  tests/test_oracle_math.py pins it
// against LAPACK (numpy.linalg.svd).
void svd3(const M3& F, M3& U, double s[3], M3& V);

// Symmetric eigen-decomposition A = Q diag(w) Q^T (cyclic Jacobi), n <= 12, w ascending.
void sym_eig(int n, const double* A, double* w, double* Q);

// IglUtils::makePD (IglUtils.hpp:119-137): clamp negative eigenvalues to zero;
// the matrix is left untouched when the smallest eigenvalue is >= 0.
void make_pd(int n, double* A);
// IglUtils::makePD2d (IglUtils.hpp:138-177), closed form, m = [a b; b d] column-major 2x2.
void make_pd2d(double* m);

} // namespace orc
