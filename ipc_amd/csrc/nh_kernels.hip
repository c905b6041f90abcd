// Per-tetrahedron neo-Hookean kernels for gfx950 (CDNA4), fp64.
//
// What the reference does per element on the CPU (tbb::parallel_for bodies):
//   energy   Energy.cpp:195-242  + NeoHookeanEnergy.cpp:55-69
//   gradient Energy.cpp:334-366  + NeoHookeanEnergy.cpp:138-153 + IglUtils.cpp:635-669
//   Hessian  Energy.cpp:368-408, 448-562 + NeoHookeanEnergy.cpp:71-136 + IglUtils.hpp:337-432
//   scatter  Energy.cpp:270-288, 317-327 -> IglUtils.hpp:39-116 -> LinSysSolver.hpp:402-410
//   step     get_feasible_steps.cpp:75-172
// Here one lane owns one tetrahedron.  Element data is SoA so a wave reads 512 contiguous
// bytes per field; nodal positions are 24-byte records gathered through L2.  The 9x9
// dP/dF of the reference (21-term sums per entry, Energy.cpp:552) is never formed: with
// beta_a = V^T (dN_a/dX) the 3x3 node-pair block is  H_ac = U T_ac U^T  where T_ac collects
// the sigma-space A block (3x3, PSD-clamped) and the three 2x2 B blocks (makePD2d-clamped)
// -- the same numbers, ~2.5 kflop instead of ~7 kflop per element.
#include "nh_kernels.h"
#include <algorithm>
#include <cfloat>
#include <cstdlib>

namespace ipcgpu {

namespace {

constexpr int BLOCK = 256;

struct d3 {
    double x, y, z;
};
__device__ __forceinline__ d3 ld3(const double* p, int v)
{
    const double* q = p + 3 * (size_t)v;
    return { q[0], q[1], q[2] };
}

__device__ __forceinline__ double det3(const double F[9])
{
    // column-major F[i + 3 j]
    return F[0] * (F[4] * F[8] - F[7] * F[5]) - F[3] * (F[1] * F[8] - F[7] * F[2]) + F[6] * (F[1] * F[5] - F[4] * F[2]);
}

// F = [x1-x0, x2-x0, x3-x0] * A   (Energy.cpp:344-355), column-major
__device__ __forceinline__ void deformation_gradient(const d3& x0, const d3& x1, const d3& x2, const d3& x3,
    const double A[9], double F[9])
{
    const double d[9] = { x1.x - x0.x, x1.y - x0.y, x1.z - x0.z, x2.x - x0.x, x2.y - x0.y, x2.z - x0.z,
        x3.x - x0.x, x3.y - x0.y, x3.z - x0.z };
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) F[i + 3 * j] = d[i] * A[0 + 3 * j] + d[i + 3] * A[1 + 3 * j] + d[i + 6] * A[2 + 3 * j];
}

__device__ __forceinline__ void load_A(const ElemView& v, int t, double A[9])
{
#pragma unroll
    for (int k = 0; k < 9; ++k) A[k] = v.A[(size_t)k * v.nT + t];
}

// One-sided Jacobi SVD  F = U diag(s) V^T with the output convention of the reference's SVD
// (ImplicitQRSVD.h:681-850): U, V rotations, |s0|>=|s1|>=|s2|, only s2 may be negative.
__device__ void svd3(const double Fin[9], double U[9], double s[3], double V[9])
{
    double G[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        G[i] = Fin[i];
        V[i] = 0.0;
    }
    V[0] = V[4] = V[8] = 1.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = (pq == 2) ? 1 : 0;
            const int q = (pq == 0) ? 1 : 2;
            double al = G[3 * p] * G[3 * p] + G[3 * p + 1] * G[3 * p + 1] + G[3 * p + 2] * G[3 * p + 2];
            double be = G[3 * q] * G[3 * q] + G[3 * q + 1] * G[3 * q + 1] + G[3 * q + 2] * G[3 * q + 2];
            double ga = G[3 * p] * G[3 * q] + G[3 * p + 1] * G[3 * q + 1] + G[3 * p + 2] * G[3 * q + 2];
            if (ga != 0.0 && fabs(ga) > 1e-16 * sqrt(al * be)) {
                rotated = true;
                double zeta = (be - al) / (2.0 * ga);
                double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = rsqrt(1.0 + t * t), sn = c * t;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    double gp = G[3 * p + i], gq = G[3 * q + i];
                    G[3 * p + i] = c * gp - sn * gq;
                    G[3 * q + i] = sn * gp + c * gq;
                    double vp = V[3 * p + i], vq = V[3 * q + i];
                    V[3 * p + i] = c * vp - sn * vq;
                    V[3 * q + i] = sn * vp + c * vq;
                }
            }
        }
        if (!rotated) break;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) s[j] = sqrt(G[3 * j] * G[3 * j] + G[3 * j + 1] * G[3 * j + 1] + G[3 * j + 2] * G[3 * j + 2]);
    // sort columns by descending singular value (3-element network)
    auto cswap = [&](int a, int b) {
        if (s[a] < s[b]) {
            double t = s[a];
            s[a] = s[b];
            s[b] = t;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                t = G[3 * a + i]; G[3 * a + i] = G[3 * b + i]; G[3 * b + i] = t;
                t = V[3 * a + i]; V[3 * a + i] = V[3 * b + i]; V[3 * b + i] = t;
            }
        }
    };
    cswap(0, 1);
    cswap(1, 2);
    cswap(0, 1);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double inv = (s[j] > 0.0) ? 1.0 / s[j] : 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) U[3 * j + i] = G[3 * j + i] * inv;
    }
    if (s[2] < 1e-14 * s[0]) { // rank deficient: complete U with the cross product (elements in this state are rejected upstream)
        U[6] = U[1] * U[5] - U[2] * U[4];
        U[7] = U[2] * U[3] - U[0] * U[5];
        U[8] = U[0] * U[4] - U[1] * U[3];
    }
    if (det3(V) < 0.0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            V[6 + i] = -V[6 + i];
            U[6 + i] = -U[6 + i];
        }
    }
    if (det3(U) < 0.0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) U[6 + i] = -U[6 + i];
        s[2] = -s[2];
    }
}

// symmetric 3x3 PSD projection = IglUtils::makePD (IglUtils.hpp:119-137); S = {a00,a11,a22,a01,a12,a02}.
// Fast exit when all leading minors are positive (then every eigenvalue is > 0 and makePD returns unchanged).
__device__ void make_pd3(double S[6])
{
    double m2 = S[0] * S[1] - S[3] * S[3];
    double d3v = S[0] * (S[1] * S[2] - S[4] * S[4]) - S[3] * (S[3] * S[2] - S[4] * S[5]) + S[5] * (S[3] * S[4] - S[1] * S[5]);
    if (S[0] > 0.0 && m2 > 0.0 && d3v > 0.0) return;
    // cyclic Jacobi eigen-decomposition
    double A[9] = { S[0], S[3], S[5], S[3], S[1], S[4], S[5], S[4], S[2] };
    double Q[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    for (int sweep = 0; sweep < 50; ++sweep) {
        double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        double dg = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
        if (off <= 1e-32 * dg || off == 0.0) break;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = (pq == 2) ? 1 : 0;
            const int q = (pq == 0) ? 1 : 2;
            double apq = A[p + 3 * q];
            if (apq == 0.0) continue;
            double theta = (A[q + 3 * q] - A[p + 3 * p]) / (2.0 * apq);
            double t = copysign(1.0, theta) / (fabs(theta) + sqrt(theta * theta + 1.0));
            double c = rsqrt(t * t + 1.0), sn = t * c;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                double akp = A[k + 3 * p], akq = A[k + 3 * q];
                A[k + 3 * p] = c * akp - sn * akq;
                A[k + 3 * q] = sn * akp + c * akq;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                double apk = A[p + 3 * k], aqk = A[q + 3 * k];
                A[p + 3 * k] = c * apk - sn * aqk;
                A[q + 3 * k] = sn * apk + c * aqk;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                double qkp = Q[k + 3 * p], qkq = Q[k + 3 * q];
                Q[k + 3 * p] = c * qkp - sn * qkq;
                Q[k + 3 * q] = sn * qkp + c * qkq;
            }
        }
    }
    double w0 = A[0], w1 = A[4], w2 = A[8];
    if (w0 >= 0.0 && w1 >= 0.0 && w2 >= 0.0) return; // smallest eigenvalue >= 0: untouched
    w0 = fmax(w0, 0.0);
    w1 = fmax(w1, 0.0);
    w2 = fmax(w2, 0.0);
    auto rec = [&](int i, int j) { return Q[i] * w0 * Q[j] + Q[i + 3] * w1 * Q[j + 3] + Q[i + 6] * w2 * Q[j + 6]; };
    S[0] = rec(0, 0);
    S[1] = rec(1, 1);
    S[2] = rec(2, 2);
    S[3] = rec(0, 1);
    S[4] = rec(1, 2);
    S[5] = rec(0, 2);
}

// IglUtils::makePD2d (IglUtils.hpp:138-177) on [[a, b],[b, d]], same operations in the same order
__device__ __forceinline__ void make_pd2d(double& m00, double& m01, double& m11)
{
    const double a = m00, b = m01, d = m11;
    const double b2 = b * b;
    const double D = a * d - b2;
    const double T_div_2 = (a + d) / 2.0;
    const double sqrtTT4D = sqrt(T_div_2 * T_div_2 - D);
    const double L2 = T_div_2 - sqrtTT4D;
    if (L2 < 0.0) {
        const double L1 = T_div_2 + sqrtTT4D;
        if (L1 <= 0.0) {
            m00 = m01 = m11 = 0.0;
        }
        else if (b2 == 0.0) {
            m00 = L1;
            m01 = m11 = 0.0;
        }
        else {
            const double L1md = L1 - d;
            const double L1md_div_L1 = L1md / L1;
            m00 = L1md_div_L1 * L1md;
            m01 = b * L1md_div_L1;
            m11 = b2 / L1;
        }
    }
}

// shape-function gradients: b_a[j] = dN_a/dX_j ; b_{k+1}[j] = A(k, j), b_0 = -sum (IglUtils.hpp:417-430)
__device__ __forceinline__ void shape_grads(const double A[9], double b[4][3])
{
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        b[1][j] = A[0 + 3 * j];
        b[2][j] = A[1 + 3 * j];
        b[3][j] = A[2 + 3 * j];
        b[0][j] = -b[1][j] - b[2][j] - b[3][j];
    }
}

// First Piola-Kirchhoff stress times w (NeoHookeanEnergy.cpp:138-153): P = mu (F - F^-T) + lam ln J F^-T
__device__ __forceinline__ void piola(const double F[9], double mu, double lam, double w, double P[9])
{
    if (mu == 0.0 && lam == 0.0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) P[i] = 0.0;
        return;
    }
    double C[9]; // cofactor (IglUtils.hpp:448-458), column-major
    C[0] = F[4] * F[8] - F[7] * F[5];
    C[3] = F[7] * F[2] - F[1] * F[8];
    C[6] = F[1] * F[5] - F[4] * F[2];
    C[1] = F[6] * F[5] - F[3] * F[8];
    C[4] = F[0] * F[8] - F[6] * F[2];
    C[7] = F[3] * F[2] - F[0] * F[5];
    C[2] = F[3] * F[7] - F[6] * F[4];
    C[5] = F[6] * F[1] - F[0] * F[7];
    C[8] = F[0] * F[4] - F[3] * F[1];
    const double J = F[0] * C[0] + F[3] * C[3] + F[6] * C[6];
    const double invJ = 1.0 / J;
    const double k = lam * log(J);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        double fit = C[i] * invJ;
        P[i] = w * (mu * (F[i] - fit) + k * fit);
    }
}

__device__ __forceinline__ bool projected_dbc(int type, int projectDBC)
{
    return type == 1 || (type == 2 && projectDBC); // Mesh.hpp:135-144
}

// ------------------------------------------------------------------------------------------------
// Newton assembly: gradient (+ Hessian) of dt^2 * elastic energy, scattered to the nodal gradient
// and the symmetric-upper CSR.  v1 scatter: hardware fp64 atomics (global_atomic_add_f64).
// SCATTER: 0 = hardware atomics (the shipped path); 1 = plain racy stores, 2 = one store per element -- both only
// to split the kernel time into arithmetic and scatter when profiling (IPCGPU_ASM_PROBE), never for results.
template <bool HESS, int SCATTER = 0>
__global__ __launch_bounds__(BLOCK) void k_assemble(ElemView v, double coef, int projectDBC, double* __restrict__ grad,
    double* __restrict__ a)
{
    const int t = v.tetBegin + blockIdx.x * BLOCK + threadIdx.x;
    if (t >= v.tetEnd) return;
    const int4 tv = v.tet[t];
    const int vid[4] = { tv.x, tv.y, tv.z, tv.w };
    const d3 x0 = ld3(v.x, vid[0]), x1 = ld3(v.x, vid[1]), x2 = ld3(v.x, vid[2]), x3 = ld3(v.x, vid[3]);
    double A[9], F[9];
    load_A(v, t, A);
    deformation_gradient(x0, x1, x2, x3, A, F);
    const double mu = v.mu[t], lam = v.lam[t];
    const double w = coef * v.vol[t];
    double b[4][3];
    shape_grads(A, b);
    int dtype[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) dtype[k] = v.dbc[vid[k]];

    if (grad) {
        double P[9];
        piola(F, mu, lam, w, P);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (projectDBC && dtype[k] != 0) continue; // Energy.cpp:284-288
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double g = P[i] * b[k][0] + P[i + 3] * b[k][1] + P[i + 6] * b[k][2];
                atomicAdd(&grad[3 * (size_t)vid[k] + i], g);
            }
        }
    }
    if (!HESS) return;
    if (mu == 0.0 && lam == 0.0) return;

    double U[9], s[3], V[9];
    svd3(F, U, s, V);
    // sigma-space derivatives (NeoHookeanEnergy.cpp:71-136)
    const double L = log(s[0] * s[1] * s[2]);
    double dE[3], inv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        inv[i] = 1.0 / s[i];
        dE[i] = mu * (s[i] - inv[i]) + lam * inv[i] * L;
    }
    double A3[6]; // {00,11,22,01,12,02}
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double inv2 = inv[i] * inv[i];
        A3[i] = mu * (1.0 + inv2) - lam * inv2 * (L - 1.0);
    }
    A3[3] = lam * inv[0] * inv[1];
    A3[4] = lam * inv[1] * inv[2];
    A3[5] = lam * inv[2] * inv[0];
    make_pd3(A3);
    const double middle = mu - lam * L;
    const double BL[3] = { (mu + middle * inv[0] * inv[1]) / 2.0, (mu + middle * inv[1] * inv[2]) / 2.0,
        (mu + middle * inv[2] * inv[0]) / 2.0 };
    // 2x2 blocks (Energy.cpp:467-491): k -> (i, j) = (k, (k+1)%3)
    double B00[3], B01[3], B11[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int kp = (k + 1) % 3;
        double rc = dE[k] + dE[kp];
        const double ss = s[k] + s[kp];
        const double eps = 1.0e-6;
        rc /= (ss < eps) ? (2.0 * eps) : (2.0 * ss);
        B00[k] = B11[k] = BL[k] + rc;
        B01[k] = BL[k] - rc;
        make_pd2d(B00[k], B01[k], B11[k]);
    }
    // M entries (Energy.cpp:497-529), scaled by w:  Bd[p][q] = M(pq,pq), Bo[p][q] = M(pq,qp)
    double Bd[3][3], Bo[3][3], Ad[3][3];
    Bd[0][1] = w * B00[0]; Bd[1][0] = w * B11[0]; Bo[0][1] = Bo[1][0] = w * B01[0];
    Bd[1][2] = w * B00[1]; Bd[2][1] = w * B11[1]; Bo[1][2] = Bo[2][1] = w * B01[1];
    Bd[0][2] = w * B11[2]; Bd[2][0] = w * B00[2]; Bo[0][2] = Bo[2][0] = w * B01[2];
    Bd[0][0] = Bd[1][1] = Bd[2][2] = 0.0;
    Bo[0][0] = Bo[1][1] = Bo[2][2] = 0.0;
    Ad[0][0] = w * A3[0]; Ad[1][1] = w * A3[1]; Ad[2][2] = w * A3[2];
    Ad[0][1] = Ad[1][0] = w * A3[3];
    Ad[1][2] = Ad[2][1] = w * A3[4];
    Ad[0][2] = Ad[2][0] = w * A3[5];
    // beta_a = V^T b_a
    double beta[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 3; ++q) beta[k][q] = V[3 * q] * b[k][0] + V[3 * q + 1] * b[k][1] + V[3 * q + 2] * b[k][2];

    bool proj[4];
    int rb[4], rl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        proj[k] = projected_dbc(dtype[k], projectDBC);
        rb[k] = v.rowBase[vid[k]];
        rl[k] = v.rowLen[vid[k]];
    }

    int e = 0;
#pragma unroll
    for (int ka = 0; ka < 4; ++ka) {
#pragma unroll
        for (int kc = ka; kc < 4; ++kc) {
            const bool diag = (ka == kc);
            int p0 = 0;
            if (!diag) {
                p0 = v.edgeP0[(size_t)e * v.nT + t];
                ++e;
            }
            if (proj[ka] || proj[kc]) continue; // rows and columns of projected nodes are dropped (IglUtils.hpp:45-53)
            // T[p][p'] in sigma space
            double T[3][3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int pp = 0; pp < 3; ++pp) {
                    if (p == pp) {
                        const int q1 = (p + 1) % 3, q2 = (p + 2) % 3;
                        T[p][p] = Ad[p][p] * beta[ka][p] * beta[kc][p] + Bd[p][q1] * beta[ka][q1] * beta[kc][q1]
                            + Bd[p][q2] * beta[ka][q2] * beta[kc][q2];
                    }
                    else {
                        T[p][pp] = Ad[p][pp] * beta[ka][p] * beta[kc][pp] + Bo[p][pp] * beta[ka][pp] * beta[kc][p];
                    }
                }
            // H = U T U^T  (U column-major: U(i,p) = U[i + 3 p])
            double UT[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int pp = 0; pp < 3; ++pp) UT[i][pp] = U[i] * T[0][pp] + U[i + 3] * T[1][pp] + U[i + 6] * T[2][pp];
            double H[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < 3; ++r) H[i][r] = UT[i][0] * U[r] + UT[i][1] * U[r + 3] + UT[i][2] * U[r + 6];
            if (SCATTER == 2) {
                double sum = 0.0;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int r = 0; r < 3; ++r) sum += H[i][r];
                if (sum == 1.2345e300) a[t] = sum;
                continue;
            }
            if (SCATTER == 1) {
                if (diag) {
                    const int base = rb[ka], Lr = rl[ka];
                    a[base + 0] = H[0][0]; a[base + 1] = H[0][1]; a[base + 2] = H[0][2];
                    a[base + Lr + 0] = H[1][1]; a[base + Lr + 1] = H[1][2]; a[base + 2 * Lr - 1] = H[2][2];
                }
                else {
                    const bool aFirst = vid[ka] < vid[kc];
                    const int Lr = aFirst ? rl[ka] : rl[kc];
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const int rowOff = (r == 0) ? 0 : (r == 1 ? (Lr - 1) : (2 * Lr - 3));
#pragma unroll
                        for (int c = 0; c < 3; ++c) a[p0 + rowOff + c] = aFirst ? H[r][c] : H[c][r];
                    }
                }
                continue;
            }
            if (diag) {
                const int base = rb[ka], Lr = rl[ka];
                atomicAdd(&a[base + 0], H[0][0]);
                atomicAdd(&a[base + 1], H[0][1]);
                atomicAdd(&a[base + 2], H[0][2]);
                atomicAdd(&a[base + Lr + 0], H[1][1]);
                atomicAdd(&a[base + Lr + 1], H[1][2]);
                atomicAdd(&a[base + 2 * Lr - 1], H[2][2]);
            }
            else {
                // block row = smaller global id; H_ac has rows of node a and columns of node c
                const bool aFirst = vid[ka] < vid[kc];
                const int Lr = aFirst ? rl[ka] : rl[kc];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int rowOff = (r == 0) ? 0 : (r == 1 ? (Lr - 1) : (2 * Lr - 3));
#pragma unroll
                    for (int c = 0; c < 3; ++c) atomicAdd(&a[p0 + rowOff + c], aFirst ? H[r][c] : H[c][r]);
                }
            }
        }
    }
}

__global__ __launch_bounds__(BLOCK) void k_node_init(ElemView v, int projectDBC, int owner, double* __restrict__ a,
    double* __restrict__ grad)
{
    const int n = blockIdx.x * BLOCK + threadIdx.x;
    if (n >= v.nV) return;
    const int type = v.dbc[n];
    const bool proj = projected_dbc(type, projectDBC);
    if (grad) {
        double g[3] = { 0, 0, 0 };
        if (!proj && owner) {
            const double m = v.mass[n];
#pragma unroll
            for (int i = 0; i < 3; ++i) g[i] = m * (v.x[3 * (size_t)n + i] - v.xTilde[3 * (size_t)n + i]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) grad[3 * (size_t)n + i] = g[i];
    }
    if (a && owner) {
        const int base = v.rowBase[n], Lr = v.rowLen[n];
        const double d = proj ? 1.0 : v.mass[n];
        a[base] = d;
        a[base + Lr] = d;
        a[base + 2 * Lr - 1] = d;
    }
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double x, double* sm)
{
    // fixed-order tree: deterministic
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sm[wv] = x;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        for (int i = 0; i < BLOCK / 64; ++i) r += sm[i];
    }
    return r;
}

__device__ __forceinline__ double nh_psi(const double F[9], double mu, double lam)
{
    if (mu == 0.0 && lam == 0.0) return 0.0;
    double I1 = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) I1 += F[i] * F[i];
    const double lJ = log(det3(F));
    return mu / 2.0 * (I1 - 3.0) - (mu - lam / 2.0 * lJ) * lJ; // NeoHookeanEnergy.cpp:64-68 with sum sigma^2 = |F|^2, prod sigma = det F
}

__global__ __launch_bounds__(BLOCK) void k_energy(ElemView v, double coef, int withInertia, int owner, double* __restrict__ partial)
{
    __shared__ double sm[BLOCK / 64];
    const int gid = blockIdx.x * BLOCK + threadIdx.x;
    double e = 0.0;
    const int t = v.tetBegin + gid;
    if (t < v.tetEnd) {
        const int4 tv = v.tet[t];
        double A[9], F[9];
        load_A(v, t, A);
        deformation_gradient(ld3(v.x, tv.x), ld3(v.x, tv.y), ld3(v.x, tv.z), ld3(v.x, tv.w), A, F);
        e = coef * v.vol[t] * nh_psi(F, v.mu[t], v.lam[t]);
    }
    if (withInertia && owner && gid < v.nV) {
        double d2 = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double d = v.x[3 * (size_t)gid + i] - v.xTilde[3 * (size_t)gid + i];
            d2 += d * d;
        }
        e += d2 * v.mass[gid] / 2.0; // Optimizer.cpp:3234
    }
    double r = block_sum(e, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

__global__ __launch_bounds__(BLOCK) void k_reduce_partials(const double* __restrict__ partial, int n, double* __restrict__ out)
{
    __shared__ double sm[BLOCK / 64];
    double x = 0.0;
    for (int i = threadIdx.x; i < n; i += BLOCK) x += partial[i];
    double r = block_sum(x, sm);
    if (threadIdx.x == 0) out[0] = r;
}

__global__ __launch_bounds__(BLOCK) void k_energy_per_elem(ElemView v, double* __restrict__ out)
{
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= v.nT) return;
    const int4 tv = v.tet[t];
    double A[9], F[9];
    load_A(v, t, A);
    deformation_gradient(ld3(v.x, tv.x), ld3(v.x, tv.y), ld3(v.x, tv.z), ld3(v.x, tv.w), A, F);
    out[t] = v.vol[t] * nh_psi(F, v.mu[t], v.lam[t]);
}

__global__ __launch_bounds__(BLOCK) void k_check_inversion(ElemView v, int* flag)
{
    const int t = v.tetBegin + blockIdx.x * BLOCK + threadIdx.x;
    if (t >= v.tetEnd) return;
    if (!(v.mu[t] != 0.0 && v.lam[t] != 0.0)) return; // Mesh.cpp:750
    const int4 tv = v.tet[t];
    const d3 x0 = ld3(v.x, tv.x), x1 = ld3(v.x, tv.y), x2 = ld3(v.x, tv.z), x3 = ld3(v.x, tv.w);
    const double e[9] = { x1.x - x0.x, x1.y - x0.y, x1.z - x0.z, x2.x - x0.x, x2.y - x0.y, x2.z - x0.z, x3.x - x0.x,
        x3.y - x0.y, x3.z - x0.z };
    if (det3(e) < 0.0) atomicOr(flag, 1);
}

// ---- tetrahedron inversion step bound (get_feasible_steps.cpp:9-172) ----------------------------
struct cplx {
    double re, im;
};
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re }; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return { a.re + b.re, a.im + b.im }; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return { a.re - b.re, a.im - b.im }; }
__device__ __forceinline__ cplx cscale(cplx a, double s) { return { a.re * s, a.im * s }; }
__device__ __forceinline__ cplx cdiv(cplx a, cplx b)
{
    double d = b.re * b.re + b.im * b.im;
    return { (a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d };
}
__device__ __forceinline__ double cabs_(cplx a) { return hypot(a.re, a.im); }
__device__ __forceinline__ cplx csqrt_(cplx z)
{
    if (z.re == 0.0 && z.im == 0.0) return { 0.0, 0.0 };
    double m = cabs_(z);
    double t = sqrt(0.5 * (m + fabs(z.re)));
    if (z.re >= 0.0) return { t, z.im / (2.0 * t) };
    return { fabs(z.im) / (2.0 * t), copysign(t, z.im) };
}
__device__ __forceinline__ cplx cpow_third(cplx z)
{
    if (z.re == 0.0 && z.im == 0.0) return { 0.0, 0.0 };
    if (z.im == 0.0 && z.re > 0.0) return { cbrt(z.re), 0.0 };
    double r = cbrt(cabs_(z)), th = atan2(z.im, z.re) / 3.0;
    return { r * cos(th), r * sin(th) };
}
__device__ double quad_root(double a, double b, double c, double tol)
{
    double t;
    if (fabs(a) <= tol) t = -c / b;
    else {
        double desc = b * b - 4 * a * c;
        if (desc > 0) {
            t = (-b - sqrt(desc)) / (2 * a);
            if (t < 0) t = (-b + sqrt(desc)) / (2 * a);
        }
        else t = -1;
    }
    return t;
}
__device__ double cubic_root(double a, double b, double c, double d, double tol)
{
    double t = -1;
    if (fabs(a) <= tol) return quad_root(b, c, d, tol);
    cplx delta0 = { b * b - 3 * a * c, 0 };
    cplx delta1 = { 2 * b * b * b - 9 * a * b * c + 27 * a * a * d, 0 };
    cplx disc = csqrt_(csub(cmul(delta1, delta1), cscale(cmul(cmul(delta0, delta0), delta0), 4.0)));
    cplx C = cpow_third(cscale(cadd(delta1, disc), 0.5));
    if (cabs_(C) == 0.0) C = cpow_third(cscale(csub(delta1, disc), 0.5));
    const double h = sqrt(3.0) / 2.0;
    cplx u2 = { -0.5, h }, u3 = { -0.5, -h };
    cplx bb = { b, 0 };
    double m3a = -3.0 * a;
    cplx t1 = cscale(cadd(cadd(bb, C), cdiv(delta0, C)), 1.0 / m3a);
    cplx u2C = cmul(u2, C), u3C = cmul(u3, C);
    cplx t2 = cscale(cadd(cadd(bb, u2C), cdiv(delta0, u2C)), 1.0 / m3a);
    cplx t3 = cscale(cadd(cadd(bb, u3C), cdiv(delta0, u3C)), 1.0 / m3a);
    if ((fabs(t1.im) < tol) && (t1.re > 0)) t = t1.re;
    if ((fabs(t2.im) < tol) && (t2.re > 0) && ((t2.re < t) || (t < 0))) t = t2.re;
    if ((fabs(t3.im) < tol) && (t3.re > 0) && ((t3.re < t) || (t < 0))) t = t3.re;
    return t;
}

__global__ __launch_bounds__(BLOCK) void k_inversion_step(ElemView v, const double* __restrict__ p, double slackness,
    unsigned long long* __restrict__ outMin)
{
    const int t = v.tetBegin + blockIdx.x * BLOCK + threadIdx.x;
    double res = 1e20;
    if (t < v.tetEnd) {
        const int4 tv = v.tet[t];
        const d3 x0 = ld3(v.x, tv.x), x1 = ld3(v.x, tv.y), x2 = ld3(v.x, tv.z), x3 = ld3(v.x, tv.w);
        const d3 p0 = ld3(p, tv.x), p1 = ld3(p, tv.y), p2 = ld3(p, tv.z), p3 = ld3(p, tv.w);
        const double v0[3] = { x1.x - x0.x, x1.y - x0.y, x1.z - x0.z }, v1[3] = { x2.x - x0.x, x2.y - x0.y, x2.z - x0.z },
                     v2[3] = { x3.x - x0.x, x3.y - x0.y, x3.z - x0.z };
        const double q0[3] = { p1.x - p0.x, p1.y - p0.y, p1.z - p0.z }, q1[3] = { p2.x - p0.x, p2.y - p0.y, p2.z - p0.z },
                     q2[3] = { p3.x - p0.x, p3.y - p0.y, p3.z - p0.z };
        auto cross = [](const double* x, const double* y, double* z) {
            z[0] = x[1] * y[2] - x[2] * y[1];
            z[1] = x[2] * y[0] - x[0] * y[2];
            z[2] = x[0] * y[1] - x[1] * y[0];
        };
        auto dot = [](const double* x, const double* y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; };
        double pxp[3], vxp[3], pxv[3], vxv[3], mix[3];
        cross(q0, q1, pxp);
        cross(v0, q1, vxp);
        cross(q0, v1, pxv);
        cross(v0, v1, vxv);
        for (int c = 0; c < 3; ++c) mix[c] = vxp[c] + pxv[c];
        const double a = dot(q2, pxp);
        const double b = dot(v2, pxp) + dot(q2, mix);
        const double c = dot(q2, vxv) + dot(v2, mix);
        const double d = (1.0 - slackness) * dot(v2, vxv);
        const double r = cubic_root(a, b, c, d, 1.0e-6);
        res = (r >= 0) ? r : 1e20;
    }
    // all results are >= 0, so their bit patterns order like unsigned integers
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) res = fmin(res, __shfl_down(res, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMin(outMin, (unsigned long long)__double_as_longlong(res));
}

// ---- nodal helpers -----------------------------------------------------------------------------
__global__ void k_step_forward(int n, const double* __restrict__ x0, const double* __restrict__ p, double alpha, double* __restrict__ x)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = x0[i] + alpha * p[i];
}
__global__ void k_max_abs(int n, const double* __restrict__ v, unsigned long long* __restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    double m = (i < n) ? fabs(v[i]) : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}
__global__ void k_fill(double* p, size_t n, double v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void k_negate(int n, const double* __restrict__ in, double* __restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = -in[i];
}
__global__ void k_colmajor_to_aos(int nV, const double* __restrict__ src, double* __restrict__ dst)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * nV) {
        int v = i / 3, c = i % 3;
        dst[i] = src[v + (size_t)nV * c];
    }
}
__global__ void k_aos_to_colmajor(int nV, const double* __restrict__ src, double* __restrict__ dst)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * nV) {
        int c = i / nV, v = i % nV;
        dst[i] = src[3 * (size_t)v + c];
    }
}
__global__ void k_csr_symv_zero(int n, double* y)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = 0.0;
}
__global__ void k_csr_symv(int nRows, const int* __restrict__ ia, const int* __restrict__ ja, const double* __restrict__ a,
    const double* __restrict__ x, double* __restrict__ y)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nRows) return;
    double acc = 0.0;
    const double xr = x[r];
    for (int k = ia[r]; k < ia[r + 1]; ++k) {
        const int c = ja[k];
        acc += a[k] * x[c];
        if (c != r) atomicAdd(&y[c], a[k] * xr);
    }
    atomicAdd(&y[r], acc);
}
__global__ void k_precond_diag(int nRows, const int* __restrict__ ia, const double* __restrict__ a, const double* __restrict__ in,
    double* __restrict__ out)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nRows) out[r] = in[r] / a[ia[r]]; // the diagonal is the first entry of every upper-CSR row
}
__global__ void k_be_update(int nV, const int* __restrict__ dbc, const double* __restrict__ x, double* __restrict__ xPrev,
    double* __restrict__ vel, double* __restrict__ xTilde, double dt, double gx, double gy, double gz)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * nV) return;
    const int v = i / 3, c = i % 3;
    const double g = (c == 0) ? gx : (c == 1 ? gy : gz);
    const double xi = x[i];
    const double vi = (xi - xPrev[i]) / dt;
    vel[i] = vi;
    xPrev[i] = xi;
    xTilde[i] = (dbc[v] != 0) ? xi : (xi + (vi * dt + dt * dt * g));
}
__global__ void k_twist_dir(int nH, const int* __restrict__ ids, const double* __restrict__ ang, double cy, double cz,
    const double* __restrict__ x, double* __restrict__ p)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nH) return;
    const int v = ids[i];
    const double cs = cos(ang[i]), sn = sin(ang[i]);
    const double y = x[3 * (size_t)v + 1] - cy, z = x[3 * (size_t)v + 2] - cz;
    p[3 * (size_t)v + 0] = 0.0;
    p[3 * (size_t)v + 1] = (cs * y - sn * z + cy) - x[3 * (size_t)v + 1];
    p[3 * (size_t)v + 2] = (sn * y + cs * z + cz) - x[3 * (size_t)v + 2];
}

inline int nblk(long long n, int b = BLOCK) { return (int)((n + b - 1) / b); }

} // namespace

void launch_node_init(const ElemView& v, int projectDBC, bool ownerRank, double* a, double* grad, hipStream_t s)
{
    if (v.nV) hipLaunchKernelGGL(k_node_init, dim3(nblk(v.nV)), dim3(BLOCK), 0, s, v, projectDBC, ownerRank ? 1 : 0, a, grad);
}

void launch_assemble(const ElemView& v, double coef, int projectDBC, double* grad, double* a, hipStream_t s)
{
    const int n = v.tetEnd - v.tetBegin;
    if (n <= 0) return;
    static const int probe = std::getenv("IPCGPU_ASM_PROBE") ? std::atoi(std::getenv("IPCGPU_ASM_PROBE")) : 0;
    if (a && probe == 1) hipLaunchKernelGGL((k_assemble<true, 1>), dim3(nblk(n)), dim3(BLOCK), 0, s, v, coef, projectDBC, grad, a);
    else if (a && probe == 2) hipLaunchKernelGGL((k_assemble<true, 2>), dim3(nblk(n)), dim3(BLOCK), 0, s, v, coef, projectDBC, grad, a);
    else if (a) hipLaunchKernelGGL((k_assemble<true, 0>), dim3(nblk(n)), dim3(BLOCK), 0, s, v, coef, projectDBC, grad, a);
    else hipLaunchKernelGGL((k_assemble<false, 0>), dim3(nblk(n)), dim3(BLOCK), 0, s, v, coef, projectDBC, grad, a);
}
const char* assemble_kernel_name() { return "k_assemble"; }

void launch_energy(const ElemView& v, double coef, bool withInertia, bool ownerRank, double* partial, int partialCap, double* out,
    hipStream_t s)
{
    const int n = std::max(v.tetEnd - v.tetBegin, withInertia ? v.nV : 0);
    const int nb = std::max(1, nblk(n));
    (void)partialCap;
    hipLaunchKernelGGL(k_energy, dim3(nb), dim3(BLOCK), 0, s, v, coef, withInertia ? 1 : 0, ownerRank ? 1 : 0, partial);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(BLOCK), 0, s, partial, nb, out);
}
void launch_energy_per_elem(const ElemView& v, double* perElem, hipStream_t s)
{
    if (v.nT) hipLaunchKernelGGL(k_energy_per_elem, dim3(nblk(v.nT)), dim3(BLOCK), 0, s, v, perElem);
}
void launch_check_inversion(const ElemView& v, int* flag, hipStream_t s)
{
    const int n = v.tetEnd - v.tetBegin;
    if (n > 0) hipLaunchKernelGGL(k_check_inversion, dim3(nblk(n)), dim3(BLOCK), 0, s, v, flag);
}
void launch_inversion_step(const ElemView& v, const double* p, double slackness, double* outMin, hipStream_t s)
{
    const int n = v.tetEnd - v.tetBegin;
    if (n > 0)
        hipLaunchKernelGGL(k_inversion_step, dim3(nblk(n)), dim3(BLOCK), 0, s, v, p, slackness, (unsigned long long*)outMin);
}
void launch_step_forward(int n3, const double* x0, const double* p, double alpha, double* x, hipStream_t s)
{
    if (n3) hipLaunchKernelGGL(k_step_forward, dim3(nblk(n3)), dim3(BLOCK), 0, s, n3, x0, p, alpha, x);
}
void launch_max_abs(int n, const double* v, double* out, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_max_abs, dim3(nblk(n)), dim3(BLOCK), 0, s, n, v, (unsigned long long*)out);
}
void launch_fill(double* p, size_t n, double v, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_fill, dim3(nblk((long long)n)), dim3(BLOCK), 0, s, p, n, v);
}
void launch_negate(int n, const double* in, double* out, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_negate, dim3(nblk(n)), dim3(BLOCK), 0, s, n, in, out);
}
void launch_colmajor_to_aos(int nV, const double* src, double* dst, hipStream_t s)
{
    if (nV) hipLaunchKernelGGL(k_colmajor_to_aos, dim3(nblk(3LL * nV)), dim3(BLOCK), 0, s, nV, src, dst);
}
void launch_aos_to_colmajor(int nV, const double* src, double* dst, hipStream_t s)
{
    if (nV) hipLaunchKernelGGL(k_aos_to_colmajor, dim3(nblk(3LL * nV)), dim3(BLOCK), 0, s, nV, src, dst);
}
void launch_csr_symv(int nRows, const int* ia, const int* ja, const double* a, const double* x, double* y, hipStream_t s)
{
    if (!nRows) return;
    hipLaunchKernelGGL(k_csr_symv_zero, dim3(nblk(nRows)), dim3(BLOCK), 0, s, nRows, y);
    hipLaunchKernelGGL(k_csr_symv, dim3(nblk(nRows)), dim3(BLOCK), 0, s, nRows, ia, ja, a, x, y);
}
void launch_precond_diag(int nRows, const int* ia, const double* a, const double* in, double* out, hipStream_t s)
{
    if (nRows) hipLaunchKernelGGL(k_precond_diag, dim3(nblk(nRows)), dim3(BLOCK), 0, s, nRows, ia, a, in, out);
}
void launch_be_update(int nV, const int* dbc, const double* x, double* xPrev, double* vel, double* xTilde, double dt, double gx,
    double gy, double gz, hipStream_t s)
{
    if (nV) hipLaunchKernelGGL(k_be_update, dim3(nblk(3LL * nV)), dim3(BLOCK), 0, s, nV, dbc, x, xPrev, vel, xTilde, dt, gx, gy, gz);
}
void launch_twist_dir(int nH, const int* ids, const double* ang, double cy, double cz, const double* x, double* p, hipStream_t s)
{
    if (nH) hipLaunchKernelGGL(k_twist_dir, dim3(nblk(nH)), dim3(BLOCK), 0, s, nH, ids, ang, cy, cz, x, p);
}

} // namespace ipcgpu
