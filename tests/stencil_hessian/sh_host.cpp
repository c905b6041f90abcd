// Host build of ipc_amd/csrc/stencil_hessian_device.h (+ jacobi9_device.h) for tests/test_stencil_hessian.py and tests/test_jacobi9.py
// (test infrastructure: the product compiles the same headers with hipcc).
#include "../../ipc_amd/csrc/stencil_hessian_device.h"
#include <cstring>

using namespace ipcgpu;

namespace {
template <int M>
int psd(double* A)
{
    double C[M * M];
    for (int j = 0; j < M; ++j)
        for (int i = 0; i <= j; ++i) C[sh::su<M>(i, j)] = 0.5 * (A[i + M * j] + A[j + M * i]);
    const int sweeps = sh::project_psd<M>(C);
    for (int j = 0; j < M; ++j)
        for (int i = 0; i < M; ++i) A[i + M * j] = C[sh::su<M>(i, j)];
    return sweeps;
}
template <int NR, int K, int L, int SD = 3 * NR>
void put(const double* C, double* A12)
{
    double B[9];
    sh::pair_block<NR, K, L, SD>(C, B);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            A12[(3 * K + r) + 12 * (3 * L + c)] = B[r + 3 * c];
            A12[(3 * L + c) + 12 * (3 * K + r)] = B[r + 3 * c];
        }
}
template <int NR>
void expand(const double* C, double* A12)
{
    put<NR, 0, 0>(C, A12);
    put<NR, 0, 1>(C, A12);
    put<NR, 1, 1>(C, A12);
    if constexpr (NR >= 2) {
        put<NR, 0, 2>(C, A12);
        put<NR, 1, 2>(C, A12);
        put<NR, 2, 2>(C, A12);
    }
    if constexpr (NR >= 3) {
        put<NR, 0, 3>(C, A12);
        put<NR, 1, 3>(C, A12);
        put<NR, 2, 3>(C, A12);
        put<NR, 3, 3>(C, A12);
    }
}
template <int KIND>
double active(const double* X12, double dHat, double weight, int project, double* A12)
{
    constexpr int NR = sh::NN_OF[KIND] - 1, M = 3 * NR;
    double X[4][3];
    std::memcpy(X, X12, sizeof(X));
    double C[M * M];
    const double d = sh::active_block<KIND>(X, dHat, weight, C);
    if (project) sh::project_psd<M>(C);
    std::memset(A12, 0, 144 * sizeof(double));
    expand<NR>(C, A12);
    return d;
}
// the same through the 9 x 9 frame the device kernel uses for every kind (leading dimension 9, rows / columns >= 3 (nn - 1) zero)
template <int KIND>
double active9(const double* X12, double dHat, double weight, double* A12)
{
    constexpr int NR = sh::NN_OF[KIND] - 1;
    double X[4][3];
    std::memcpy(X, X12, sizeof(X));
    double C[81];
    const double d = sh::active_block<KIND, 9>(X, dHat, weight, C);
    sh::project_psd<9>(C);
    std::memset(A12, 0, 144 * sizeof(double));
    put<NR, 0, 0, 9>(C, A12);
    put<NR, 0, 1, 9>(C, A12);
    put<NR, 1, 1, 9>(C, A12);
    if constexpr (NR >= 2) {
        put<NR, 0, 2, 9>(C, A12);
        put<NR, 1, 2, 9>(C, A12);
        put<NR, 2, 2, 9>(C, A12);
    }
    if constexpr (NR >= 3) {
        put<NR, 0, 3, 9>(C, A12);
        put<NR, 1, 3, 9>(C, A12);
        put<NR, 2, 3, 9>(C, A12);
        put<NR, 3, 3, 9>(C, A12);
    }
    return d;
}
template <int KIND>
void para(const double* XE12, const double* sel16, double dHat, double kappa, double eps_x, int project, double* A12)
{
    double XE[4][3], sel[4][4];
    std::memcpy(XE, XE12, sizeof(XE));
    std::memcpy(sel, sel16, sizeof(sel));
    double C[81];
    sh::para_block<KIND>(XE, sel, dHat, kappa, eps_x, C);
    if (project) sh::project_psd<9>(C);
    std::memset(A12, 0, 144 * sizeof(double));
    expand<3>(C, A12);
}
} // namespace

extern "C" {
// A: M x M column-major, overwritten by V max(lambda, 0) V^T; returns the sweeps
int sh_project_psd(int M, double* A) { return M == 3 ? psd<3>(A) : (M == 6 ? psd<6>(A) : psd<9>(A)); }
// node-space 12 x 12 (column-major, ld 12) block of an active stencil: kind 0..3, X = 4 x 3 row-major node positions; returns d
double sh_active(int kind, const double* X12, double dHat, double weight, int project, double* A12)
{
    switch (kind) {
    case 0: return active<0>(X12, dHat, weight, project, A12);
    case 1: return active<1>(X12, dHat, weight, project, A12);
    case 2: return active<2>(X12, dHat, weight, project, A12);
    default: return active<3>(X12, dHat, weight, project, A12);
    }
}
double sh_active9(int kind, const double* X12, double dHat, double weight, double* A12)
{
    switch (kind) {
    case 0: return active9<0>(X12, dHat, weight, A12);
    case 1: return active9<1>(X12, dHat, weight, A12);
    case 2: return active9<2>(X12, dHat, weight, A12);
    default: return active9<3>(X12, dHat, weight, A12);
    }
}
// mollified stencil on the four edge nodes XE; sel[k][q] = 1 where node k of the distance stencil (of `kind`) is edge node q
void sh_para(int kind, const double* XE12, const double* sel16, double dHat, double kappa, double eps_x, int project, double* A12)
{
    switch (kind) {
    case 0: para<0>(XE12, sel16, dHat, kappa, eps_x, project, A12); break;
    case 1: para<1>(XE12, sel16, dHat, kappa, eps_x, project, A12); break;
    case 2: para<2>(XE12, sel16, dHat, kappa, eps_x, project, A12); break;
    default: para<3>(XE12, sel16, dHat, kappa, eps_x, project, A12); break;
    }
}
}
