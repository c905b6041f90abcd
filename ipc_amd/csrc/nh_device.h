// Device-side building blocks of the neo-Hookean element computation (gfx950, fp64), shared by the
// tet-parallel atomic kernels (nh_kernels.hip) and the patch-parallel assembly kernel.
// One lane owns one tetrahedron; see nh_kernels.hip for the reference line map.
#pragma once
#include "nh_kernels.h"

namespace ipcgpu {
namespace dev {

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

// Reciprocal / reciprocal square root / square root from the hardware seeds (v_rcp_f64, v_rsq_f64) plus two Newton
// steps: ~10 dependent instructions instead of the ~35 of the IEEE-exact expansions.  Used only inside the Jacobi
// iterations (SVD, 3x3 eigen-solve), whose fixed point does not depend on the last bit of an individual rotation.
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
}
__device__ __forceinline__ double fast_rsqrt(double x)
{
    double r = __builtin_amdgcn_rsq(x);
    double e = fma(-x * r, r, 1.0);
    r = fma(0.5 * r, e, r);
    e = fma(-x * r, r, 1.0);
    return fma(0.5 * r, e, r);
}
__device__ __forceinline__ double fast_sqrt(double x)
{
    if (!(x > 0.0)) return 0.0;
    const double r = fast_rsqrt(x);
    const double s = x * r;
    return fma(fma(-s, s, x), 0.5 * r, s);
}

// One-sided Jacobi SVD  F = U diag(s) V^T with the output convention of the reference's SVD
// (ImplicitQRSVD.h:681-850): U, V rotations, |s0|>=|s1|>=|s2|, only s2 may be negative.
__device__ inline void svd3(const double Fin[9], double U[9], double s[3], double V[9])
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
            if (ga != 0.0 && ga * ga > 1e-30 * (al * be)) { // below ~1e-15 relative the "rotation" is rounding noise: 1e-32 made 2 % of the lanes (hence most waves) run all 30 sweeps
                rotated = true;
                // (round 4: a cheaper route to the same rotation -- cos 2 theta = |d| / r, sin 2 theta = sign(d) h / r with d = be - al, h = 2 ga, two reciprocal
                // square roots instead of a reciprocal, a square root, a reciprocal and a reciprocal square root -- was tried and withdrawn: it changes the last
                // bits of U and V, which is harmless everywhere except at F = I up to round-off, where the sigma-space projection is decided by exactly those
                // bits; three from-rest scene fixtures (`seg_bed_squash`, `script_toggle_top`, `dbc_time_range`) then take other Newton paths than the CPU
                // restatement, which uses this textbook form.  It bought 1.5 us of 46.)
                const double zeta = (be - al) * (0.5 * fast_rcp(ga));
                const double t = copysign(fast_rcp(fabs(zeta) + fast_sqrt(1.0 + zeta * zeta)), zeta);
                const double c = fast_rsqrt(1.0 + t * t), sn = c * t;
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
    for (int j = 0; j < 3; ++j) s[j] = fast_sqrt(G[3 * j] * G[3 * j] + G[3 * j + 1] * G[3 * j + 1] + G[3 * j + 2] * G[3 * j + 2]);
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
        double inv = (s[j] > 0.0) ? fast_rcp(s[j]) : 0.0;
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
__device__ inline void make_pd3(double S[6])
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
        if (off <= 1e-30 * dg || off == 0.0) break;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = (pq == 2) ? 1 : 0;
            const int q = (pq == 0) ? 1 : 2;
            double apq = A[p + 3 * q];
            if (apq == 0.0) continue;
            const double theta = (A[q + 3 * q] - A[p + 3 * p]) * (0.5 * fast_rcp(apq));
            const double t = copysign(fast_rcp(fabs(theta) + fast_sqrt(theta * theta + 1.0)), theta);
            const double c = fast_rsqrt(t * t + 1.0), sn = t * c;
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
    const double disc = T_div_2 * T_div_2 - D;
    if (!(disc >= 0.0)) return; // the reference's sqrt gives NaN here and every comparison below is then false
    const double sqrtTT4D = fast_sqrt(disc);
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
// Returns ln J so that the sigma-space derivatives of the same element do not evaluate a second logarithm.
__device__ __forceinline__ double piola(const double F[9], double mu, double lam, double w, double P[9])
{
    if (mu == 0.0 && lam == 0.0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) P[i] = 0.0;
        return 0.0;
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
    const double invJ = fast_rcp(J);
    const double lnJ = log(J);
    const double k = lam * lnJ;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        double fit = C[i] * invJ;
        P[i] = w * (mu * (F[i] - fit) + k * fit);
    }
    return lnJ;
}

// Fixed corotated (FixedCoRotEnergy.cpp:62-153) in sigma space
__device__ __forceinline__ double fcr_psi(const double s[3], double mu, double lam)
{
    const double d0 = s[0] - 1.0, d1 = s[1] - 1.0, d2 = s[2] - 1.0;
    const double pm1 = s[0] * s[1] * s[2] - 1.0;
    return mu * (d0 * d0 + d1 * d1 + d2 * d2) + lam / 2.0 * pm1 * pm1; // :62-70
}
// P w = w (2 mu (F - U V^T) + lam (prod sigma - 1) cof F)  (:145-153)
__device__ __forceinline__ void fcr_piola(const double F[9], const double U[9], const double s[3], const double V[9], double mu, double lam,
    double w, double P[9])
{
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
    const double k = lam * (s[0] * s[1] * s[2] - 1.0);
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double R = U[i] * V[j] + U[i + 3] * V[j + 3] + U[i + 6] * V[j + 6];
            P[i + 3 * j] = w * (mu * 2.0 * (F[i + 3 * j] - R) + k * C[i + 3 * j]);
        }
}

__device__ __forceinline__ bool projected_dbc(int type, int projectDBC)
{
    return type == 1 || (type == 2 && projectDBC); // Mesh.hpp:135-144
}

// ------------------------------------------------------------------------------------------------
// The element computation is split in two so that kernels can place a workgroup-level exchange between them:
//   element_generators  F, P (-> nodal forces), SVD, sigma-space derivatives with their PSD clamps; the result is
//                       33 doubles per element: U (9), beta_a = V^T dN_a/dX for a = 1..3 (9; beta_0 = -sum), and the
//                       non-zero entries of the clamped sigma-space matrix M scaled by w (Ad 6, Bd 6, Bo 3)
//   pair_block          the 3x3 block H_ac = U T_ac U^T of one node pair from those generators
// The 9x9 dP/dF of the reference (21-term sums per entry, Energy.cpp:552) is never formed:
//   T_ac[p][p ] = Ad[p][p] ba_p bc_p + sum_{q != p} Bd[p][q] ba_q bc_q          (M(pq,pq) entries)
//   T_ac[p][p'] = Ad[p][p'] ba_p bc_p' + Bo[p][p'] ba_p' bc_p                    (M(pp,p'p') and M(pp',p'p))
struct ElemGen {
    double U[9];
    double beta[4][3];
    double Ad[6]; // {00,11,22,01,12,02}
    double Bd[6]; // {01,10,12,21,02,20}
    double Bo[3]; // {01,12,02}
    int vid[4];
    int dtype[4];
    bool active; // false for zero-stiffness (kinematic) elements
};

template <class GradFn>
__device__ __forceinline__ void element_generators(const ElemView& v, int t, double coef, int projectDBC, bool wantGrad, bool wantHess,
    GradFn&& gradFn, ElemGen& g)
{
    const int4 tv = v.tet[t];
    g.vid[0] = tv.x; g.vid[1] = tv.y; g.vid[2] = tv.z; g.vid[3] = tv.w;
    const d3 x0 = ld3(v.x, tv.x), x1 = ld3(v.x, tv.y), x2 = ld3(v.x, tv.z), x3 = ld3(v.x, tv.w);
    double A[9], F[9];
    load_A(v, t, A);
    deformation_gradient(x0, x1, x2, x3, A, F);
    const double mu = v.mu[t], lam = v.lam[t];
    const double w = coef * v.vol[t];
    double b[4][3];
    shape_grads(A, b);
#pragma unroll
    for (int k = 0; k < 4; ++k) g.dtype[k] = v.dbc[g.vid[k]];
    const bool fcr = v.energyType == 1;
    const bool stiff = !(mu == 0.0 && lam == 0.0);
    g.active = wantHess && stiff;
    auto emitForces = [&](const double P[9]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (projectDBC && g.dtype[k] != 0) continue; // Energy.cpp:284-288
#pragma unroll
            for (int i = 0; i < 3; ++i) gradFn(k, i, P[i] * b[k][0] + P[i + 3] * b[k][1] + P[i + 6] * b[k][2]);
        }
    };
    // The nodal forces are emitted LAST (end of this function): a kernel that accumulates them in a fixed order (patch_assembly.hip
    // takes its waves one after the other) then holds the twelve values in registers for a few instructions only, not across the SVD.
    const bool fcrStress = fcr && stiff && wantGrad; // FCR needs R = U V^T for the stress
    double s[3], V[9];
    if (g.active || fcrStress) svd3(F, g.U, s, V);
    auto forces = [&]() {
        if (!wantGrad) return;
        double P[9];
        if (fcrStress) fcr_piola(F, g.U, s, V, mu, lam, w, P);
        else piola(F, mu, lam, w, P);
        emitForces(P);
    };
    if (!g.active) {
        forces();
        return;
    }
    double dE[3], inv[3], A3[6], BL[3];
    if (fcr) { // FixedCoRotEnergy.cpp:72-144
        const double prod = s[0] * s[1] * s[2];
        const double noI[3] = { s[1] * s[2], s[2] * s[0], s[0] * s[1] };
        const double kl = lam * (prod - 1.0), twoMu = mu * 2.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            dE[i] = twoMu * (s[i] - 1.0) + noI[i] * kl;
            A3[i] = twoMu + lam * noI[i] * noI[i];
        }
        A3[3] = lam * (s[2] * (prod - 1.0) + noI[0] * noI[1]);
        A3[4] = lam * (s[0] * (prod - 1.0) + noI[2] * noI[1]);
        A3[5] = lam * (s[1] * (prod - 1.0) + noI[0] * noI[2]);
        const double hl = lam / 2.0;
        BL[0] = mu - hl * s[2] * (prod - 1.0);
        BL[1] = mu - hl * s[0] * (prod - 1.0);
        BL[2] = mu - hl * s[1] * (prod - 1.0);
    }
    else { // sigma-space derivatives (NeoHookeanEnergy.cpp:71-136)
        const double L = log(s[0] * s[1] * s[2]); // det F = s0 s1 s2 (U, V rotations)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            inv[i] = fast_rcp(s[i]);
            dE[i] = mu * (s[i] - inv[i]) + lam * inv[i] * L;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double inv2 = inv[i] * inv[i];
            A3[i] = mu * (1.0 + inv2) - lam * inv2 * (L - 1.0);
        }
        A3[3] = lam * inv[0] * inv[1];
        A3[4] = lam * inv[1] * inv[2];
        A3[5] = lam * inv[2] * inv[0];
        const double middle = mu - lam * L;
        BL[0] = (mu + middle * inv[0] * inv[1]) / 2.0;
        BL[1] = (mu + middle * inv[1] * inv[2]) / 2.0;
        BL[2] = (mu + middle * inv[2] * inv[0]) / 2.0;
    }
    make_pd3(A3); // Energy.cpp:459-465
    // 2x2 blocks (Energy.cpp:467-491): k -> (i, j) = (k, (k+1)%3)
    double B00[3], B01[3], B11[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int kp = (k + 1) % 3;
        double rc = dE[k] + dE[kp];
        const double ss = s[k] + s[kp];
        const double eps = 1.0e-6;
        rc *= fast_rcp((ss < eps) ? (2.0 * eps) : (2.0 * ss));
        B00[k] = B11[k] = BL[k] + rc;
        B01[k] = BL[k] - rc;
        make_pd2d(B00[k], B01[k], B11[k]);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) g.Ad[i] = w * A3[i];
    // M entries (Energy.cpp:497-529):  Bd[pq] = M(pq,pq), Bo[pq] = M(pq,qp)
    g.Bd[0] = w * B00[0]; g.Bd[1] = w * B11[0]; g.Bo[0] = w * B01[0]; // B01: M(1,1), M(3,3), M(1,3)
    g.Bd[2] = w * B00[1]; g.Bd[3] = w * B11[1]; g.Bo[1] = w * B01[1]; // B12: M(5,5), M(7,7), M(5,7)
    g.Bd[4] = w * B11[2]; g.Bd[5] = w * B00[2]; g.Bo[2] = w * B01[2]; // B20: M(2,2), M(6,6), M(2,6)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 3; ++q) g.beta[k][q] = V[3 * q] * b[k][0] + V[3 * q + 1] * b[k][1] + V[3 * q + 2] * b[k][2];
    forces();
}

// H = U T U^T for the node pair with sigma-space shape vectors ba, bc
__device__ __forceinline__ void pair_block(const double U[9], const double ba[3], const double bc[3], const double Ad[6], const double Bd[6],
    const double Bo[3], double H[3][3])
{
    const double pi0 = ba[0] * bc[0], pi1 = ba[1] * bc[1], pi2 = ba[2] * bc[2];
    double T[3][3];
    T[0][0] = Ad[0] * pi0 + Bd[0] * pi1 + Bd[4] * pi2; // Bd[0][1], Bd[0][2]
    T[1][1] = Ad[1] * pi1 + Bd[2] * pi2 + Bd[1] * pi0; // Bd[1][2], Bd[1][0]
    T[2][2] = Ad[2] * pi2 + Bd[5] * pi0 + Bd[3] * pi1; // Bd[2][0], Bd[2][1]
    T[0][1] = Ad[3] * ba[0] * bc[1] + Bo[0] * ba[1] * bc[0];
    T[1][0] = Ad[3] * ba[1] * bc[0] + Bo[0] * ba[0] * bc[1];
    T[1][2] = Ad[4] * ba[1] * bc[2] + Bo[1] * ba[2] * bc[1];
    T[2][1] = Ad[4] * ba[2] * bc[1] + Bo[1] * ba[1] * bc[2];
    T[0][2] = Ad[5] * ba[0] * bc[2] + Bo[2] * ba[2] * bc[0];
    T[2][0] = Ad[5] * ba[2] * bc[0] + Bo[2] * ba[0] * bc[2];
    double UT[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) UT[i][pp] = U[i] * T[0][pp] + U[i + 3] * T[1][pp] + U[i + 6] * T[2][pp];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 3; ++r) H[i][r] = UT[i][0] * U[r] + UT[i][1] * U[r + 3] + UT[i][2] * U[r + 6];
}

// Gradient (+ PSD-projected Hessian) of one element, handed block by block to a Sink:
//   sink.grad(k, i, g)                      component i of the force on local node k
//   sink.wantPair(ka, kc, e)                does anybody consume block (ka, kc)?  e = local edge id (-1: diagonal)
//   sink.diagBlock(ka, H)                   3x3 symmetric block of node ka (upper triangle is used)
//   sink.offBlock(e, ka, kc, aFirst, H)     3x3 block rows ka / cols kc; aFirst <=> global id(ka) < global id(kc)
template <bool HESS, class Sink>
__device__ __forceinline__ void assemble_element(const ElemView& v, int t, double coef, int projectDBC, bool wantGrad, Sink& sink)
{
    ElemGen g;
    element_generators(v, t, coef, projectDBC, wantGrad, HESS, [&](int k, int i, double val) { sink.grad(k, i, val); }, g);
    if (!HESS || !g.active) return;
    bool proj[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) proj[k] = projected_dbc(g.dtype[k], projectDBC);
    int e = 0;
#pragma unroll
    for (int ka = 0; ka < 4; ++ka) {
#pragma unroll
        for (int kc = ka; kc < 4; ++kc) {
            const bool diag = (ka == kc);
            const int edge = diag ? -1 : e;
            if (!diag) ++e;
            if (proj[ka] || proj[kc]) continue; // rows and columns of projected nodes are dropped (IglUtils.hpp:45-53)
            if (!sink.wantPair(ka, kc, edge)) continue;
            double H[3][3];
            pair_block(g.U, g.beta[ka], g.beta[kc], g.Ad, g.Bd, g.Bo, H);
            if (diag) sink.diagBlock(ka, H);
            else sink.offBlock(edge, ka, kc, g.vid[ka] < g.vid[kc], H);
        }
    }
}

} // namespace dev
} // namespace ipcgpu
