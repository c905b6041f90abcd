// Device-side contact math (gfx950, fp64, compiled with -ffp-contract=off: the closest-feature typing below
// contains exact comparisons).
//   squared distances PP / PE / PT / EE          MeshCollisionUtils.hpp:156-161, 227-233, 685-694, 1287-1296
//   their gradients / Hessians                   (reference: MATLAB-generated g_* / H_*, :163-2015) -- here derived in
//                                                vector form: d = s^2/q with s = w.(e x f), q = |e x f|^2 (PT, EE),
//                                                d = q/|f-e|^2 (PE); chain rule through the +-1 node maps
//   closest-feature typing dType_PT / dType_EE   :2160-2210, 2073-2158
//   C2 clamped log barrier                        BarrierFunctions.hpp:56-83
//   parallel-edge mollifier                       MeshCollisionUtils.hpp:2834-2866
//   (the barrier HESSIAN of a stencil and its PSD projection: stencil_hessian_device.h, jacobi9_device.h)
#pragma once
#include <hip/hip_runtime.h>

namespace ipcgpu {
namespace cdev {

enum { K_PP = 0, K_PE = 1, K_PT = 2, K_EE = 3 };

__device__ __forceinline__ void cross3(const double* a, const double* b, double* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void sub3(const double* a, const double* b, double* c)
{
    c[0] = a[0] - b[0];
    c[1] = a[1] - b[1];
    c[2] = a[2] - b[2];
}

// ---- values only (constraint-set construction, energy, CCD) ---------------------------------------------------
__device__ __forceinline__ double d_PP(const double* v0, const double* v1)
{
    double r[3];
    sub3(v0, v1, r);
    return dot3(r, r);
}
__device__ __forceinline__ double d_PE(const double* v0, const double* v1, const double* v2)
{
    double e[3], f[3], g[3], n[3];
    sub3(v1, v0, e);
    sub3(v2, v0, f);
    sub3(f, e, g);
    cross3(e, f, n);
    return dot3(n, n) / dot3(g, g);
}
__device__ __forceinline__ double d_PT(const double* v0, const double* v1, const double* v2, const double* v3)
{
    double w[3], e[3], f[3], n[3];
    sub3(v0, v1, w);
    sub3(v2, v1, e);
    sub3(v3, v1, f);
    cross3(e, f, n);
    const double s = dot3(w, n);
    return s * s / dot3(n, n);
}
__device__ __forceinline__ double d_EE(const double* v0, const double* v1, const double* v2, const double* v3)
{
    double w[3], e[3], f[3], n[3];
    sub3(v2, v0, w);
    sub3(v1, v0, e);
    sub3(v3, v2, f);
    cross3(e, f, n);
    const double s = dot3(w, n);
    return s * s / dot3(n, n);
}
__device__ __forceinline__ double cross_sqnorm(const double* v0, const double* v1, const double* v2, const double* v3)
{
    double e[3], f[3], n[3];
    sub3(v1, v0, e);
    sub3(v3, v2, f);
    cross3(e, f, n);
    return dot3(n, n);
}

// 2 x 2 normal equations of dType_PT, solved the way the reference solves them: Eigen's pivoted LDL^T
// ((basis * basis.transpose()).ldlt().solve(...), MeshCollisionUtils.hpp:2174, 2182, 2190) restated step by step -- largest
// diagonal entry first (first maximum wins a tie), L10 = m01 / D0, D1 = m' - L10 (D0 L10), then P, L^-1, D^-1 (entries of D not
// above 1 / DBL_MAX give 0), L^-T, P^T.  Cramer's rule gives the same parameters up to rounding, but the classification
// compares them with 0 and 1 exactly, so the rounding is part of the contract.
__device__ __forceinline__ void edge_frame_param(const double* e, const double* nVec, const double* r, double* p0, double* p1)
{
    double b1[3];
    cross3(e, nVec, b1);
    const double m00 = dot3(e, e), m01 = dot3(e, b1), m11 = dot3(b1, b1);
    const double r0 = dot3(e, r), r1 = dot3(b1, r);
    double d0 = m00, d1 = m11;
    const bool swapped = fabs(m11) > fabs(m00);
    if (swapped) {
        d0 = m11;
        d1 = m00;
    }
    double x0 = swapped ? r1 : r0, x1 = swapped ? r0 : r1;
    if (!(fabs(d0) > 0.0)) { // the whole matrix is zero: Eigen leaves L = I, D = 0 and solve() returns zeros
        *p0 = 0.0;
        *p1 = 0.0;
        return;
    }
    const double l10 = m01 / d0;
    d1 -= l10 * (d0 * l10);
    x1 -= l10 * x0;
    const double tol = 1.0 / 1.7976931348623157e308;
    x0 = (fabs(d0) > tol) ? x0 / d0 : 0.0;
    x1 = (fabs(d1) > tol) ? x1 / d1 : 0.0;
    x0 -= l10 * x1;
    *p0 = swapped ? x1 : x0;
    *p1 = swapped ? x0 : x1;
}

__device__ inline int dType_PT(const double* v0, const double* v1, const double* v2, const double* v3)
{
    double e0[3], e1[3], nVec[3], r[3], e[3];
    sub3(v2, v1, e0);
    sub3(v3, v1, e1);
    cross3(e0, e1, nVec);
    double p00, p10, p01, p11, p02, p12;
    sub3(v0, v1, r);
    edge_frame_param(e0, nVec, r, &p00, &p10);
    if (p00 > 0.0 && p00 < 1.0 && p10 >= 0.0) return 3;
    sub3(v3, v2, e);
    sub3(v0, v2, r);
    edge_frame_param(e, nVec, r, &p01, &p11);
    if (p01 > 0.0 && p01 < 1.0 && p11 >= 0.0) return 4;
    sub3(v1, v3, e);
    sub3(v0, v3, r);
    edge_frame_param(e, nVec, r, &p02, &p12);
    if (p02 > 0.0 && p02 < 1.0 && p12 >= 0.0) return 5;
    if (p00 <= 0.0 && p02 >= 1.0) return 0;
    if (p01 <= 0.0 && p00 >= 1.0) return 1;
    if (p02 <= 0.0 && p01 >= 1.0) return 2;
    return 6;
}

__device__ inline int dType_EE(const double* v0, const double* v1, const double* v2, const double* v3)
{
    double u[3], v[3], w[3];
    sub3(v1, v0, u);
    sub3(v3, v2, v);
    sub3(v0, v2, w);
    const double a = dot3(u, u), b = dot3(u, v), c = dot3(v, v), d = dot3(u, w), e = dot3(v, w);
    const double D = a * c - b * b;
    double tD = D, sN, tN;
    int defaultCase = 8;
    sN = (b * e - c * d);
    if (sN <= 0.0) {
        tN = e;
        tD = c;
        defaultCase = 2;
    }
    else if (sN >= D) {
        tN = e + b;
        tD = c;
        defaultCase = 5;
    }
    else {
        tN = (a * e - b * d);
        double uxv[3];
        cross3(u, v, uxv);
        if (tN > 0.0 && tN < tD && (dot3(uxv, w) == 0.0 || dot3(uxv, uxv) < 1.0e-20 * a * c)) {
            if (sN < D / 2) {
                tN = e;
                tD = c;
                defaultCase = 2;
            }
            else {
                tN = e + b;
                tD = c;
                defaultCase = 5;
            }
        }
    }
    if (tN <= 0.0) {
        if (-d <= 0.0) return 0;
        else if (-d >= a) return 3;
        else return 6;
    }
    else if (tN >= tD) {
        if ((-d + b) <= 0.0) return 1;
        else if ((-d + b) >= a) return 4;
        else return 7;
    }
    return defaultCase;
}

// unclassified distances: type first, then the matching closed form (computePointTriD / computeEdgeEdgeD,
// MeshCollisionUtils.hpp:2255-2383)
__device__ inline double dist2_PT(const double* p, const double* a, const double* b, const double* c)
{
    switch (dType_PT(p, a, b, c)) {
    case 0: return d_PP(p, a);
    case 1: return d_PP(p, b);
    case 2: return d_PP(p, c);
    case 3: return d_PE(p, a, b);
    case 4: return d_PE(p, b, c);
    case 5: return d_PE(p, c, a);
    default: return d_PT(p, a, b, c);
    }
}
__device__ inline double dist2_EE(const double* a0, const double* a1, const double* b0, const double* b1)
{
    switch (dType_EE(a0, a1, b0, b1)) {
    case 0: return d_PP(a0, b0);
    case 1: return d_PP(a0, b1);
    case 2: return d_PE(a0, b0, b1);
    case 3: return d_PP(a1, b0);
    case 4: return d_PP(a1, b1);
    case 5: return d_PE(a1, b0, b1);
    case 6: return d_PE(b0, a0, a1);
    case 7: return d_PE(b1, a0, a1);
    default: return d_EE(a0, a1, b0, b1);
    }
}

__device__ __forceinline__ void barrier(double d, double dHat, double* b, double* gb, double* Hb)
{
    const double t2 = d - dHat, lg = log(d / dHat);
    *b = -t2 * t2 * lg;
    *gb = t2 * lg * -2.0 - (t2 * t2) / d;
    *Hb = (lg * -2.0 - t2 * 4.0 / d) + 1.0 / (d * d) * (t2 * t2);
}
__device__ __forceinline__ void mollifier(double c, double eps_x, double* e, double* eg, double* eH)
{
    if (c < eps_x) {
        const double r = c / eps_x;
        *e = (-r + 2.0) * r;
        *eg = 2.0 * (1.0 / eps_x) * (-(1.0 / eps_x) * c + 1.0);
        *eH = -2.0 / (eps_x * eps_x);
    }
    else {
        *e = 1.0;
        *eg = 0.0;
        *eH = 0.0;
    }
}

// ---- derivatives -------------------------------------------------------------------------------------------
// q(e,f) = |e x f|^2 in the 9-space (w,e,f): gradient gq[9] (w part zero) and Hessian Hq[81] column-major
__device__ inline void q_derivs(const double* e, const double* f, double* q, double* gq, double* Hq)
{
    double n[3], fxn[3], nxe[3];
    cross3(e, f, n);
    *q = dot3(n, n);
    for (int i = 0; i < 9; ++i) gq[i] = 0.0;
    for (int i = 0; i < 81; ++i) Hq[i] = 0.0;
    cross3(f, n, fxn);
    cross3(n, e, nxe);
    for (int i = 0; i < 3; ++i) {
        gq[3 + i] = 2 * fxn[i];
        gq[6 + i] = 2 * nxe[i];
    }
    const double ee = dot3(e, e), ff = dot3(f, f), ef = dot3(e, f);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double dij = (i == j) ? 1.0 : 0.0;
            Hq[(3 + i) + 9 * (3 + j)] = 2 * ff * dij - 2 * f[i] * f[j];
            Hq[(6 + i) + 9 * (6 + j)] = 2 * ee * dij - 2 * e[i] * e[j];
            const double v = 4 * e[i] * f[j] - 2 * f[i] * e[j] - 2 * ef * dij;
            Hq[(3 + i) + 9 * (6 + j)] = v;
            Hq[(6 + j) + 9 * (3 + i)] = v;
        }
}
__device__ __forceinline__ void add_skew(double* H9, int r, int c, const double* a, double sgn)
{
    H9[(3 * r + 0) + 9 * (3 * c + 1)] += -sgn * a[2];
    H9[(3 * r + 0) + 9 * (3 * c + 2)] += sgn * a[1];
    H9[(3 * r + 1) + 9 * (3 * c + 0)] += sgn * a[2];
    H9[(3 * r + 1) + 9 * (3 * c + 2)] += -sgn * a[0];
    H9[(3 * r + 2) + 9 * (3 * c + 0)] += -sgn * a[1];
    H9[(3 * r + 2) + 9 * (3 * c + 1)] += sgn * a[0];
}
// chain rule to node coordinates; coef[u * 4 + k] in {-1,0,1}; g has 12 entries, H is 12 x 12 column-major
__device__ inline void expand(int nNodes, const int* coef, const double* G9, const double* H81, double* g, double* H)
{
    for (int i = 0; i < 12; ++i) g[i] = 0.0;
    for (int k = 0; k < nNodes; ++k)
        for (int i = 0; i < 3; ++i) {
            double s = 0.0;
            for (int u = 0; u < 3; ++u) s += coef[u * 4 + k] * G9[3 * u + i];
            g[3 * k + i] = s;
        }
    if (!H) return;
    for (int i = 0; i < 144; ++i) H[i] = 0.0;
    for (int k = 0; k < nNodes; ++k)
        for (int l = 0; l < nNodes; ++l)
            for (int u = 0; u < 3; ++u) {
                if (!coef[u * 4 + k]) continue;
                for (int v = 0; v < 3; ++v) {
                    if (!coef[v * 4 + l]) continue;
                    const double c = coef[u * 4 + k] * coef[v * 4 + l];
                    for (int i = 0; i < 3; ++i)
                        for (int j = 0; j < 3; ++j) H[(3 * k + i) + 12 * (3 * l + j)] += c * H81[(3 * u + i) + 9 * (3 * v + j)];
                }
            }
}

// squared distance with gradient (12) and, when H != nullptr, Hessian (12 x 12) of stencil `kind` on X[4][3]
__device__ inline double stencil_distance(int kind, const double (*X)[3], double* g, double* H)
{
    if (kind == K_PP) {
        double r[3];
        sub3(X[0], X[1], r);
        for (int i = 0; i < 12; ++i) g[i] = 0.0;
        for (int i = 0; i < 3; ++i) {
            g[i] = 2 * r[i];
            g[3 + i] = -2 * r[i];
        }
        if (H) {
            for (int i = 0; i < 144; ++i) H[i] = 0.0;
            for (int i = 0; i < 3; ++i) {
                H[i + 12 * i] = H[(3 + i) + 12 * (3 + i)] = 2.0;
                H[i + 12 * (3 + i)] = H[(3 + i) + 12 * i] = -2.0;
            }
        }
        return dot3(r, r);
    }
    double G9[9], H81[81], gq[9], Hq[81], q;
    if (kind == K_PE) {
        double e[3], f[3], gd[3];
        sub3(X[1], X[0], e);
        sub3(X[2], X[0], f);
        sub3(f, e, gd);
        q_derivs(e, f, &q, gq, Hq);
        const double r = dot3(gd, gd);
        const double gr[9] = { 0, 0, 0, -2 * gd[0], -2 * gd[1], -2 * gd[2], 2 * gd[0], 2 * gd[1], 2 * gd[2] };
        for (int i = 0; i < 9; ++i) G9[i] = gq[i] / r - (q / (r * r)) * gr[i];
        if (H)
            for (int i = 0; i < 9; ++i)
                for (int j = 0; j < 9; ++j) {
                    double Hr = 0.0;
                    if (i >= 3 && j >= 3) {
                        const int bi = (i - 3) / 3, bj = (j - 3) / 3, ci = (i - 3) % 3, cj = (j - 3) % 3;
                        if (ci == cj) Hr = (bi == bj) ? 2.0 : -2.0;
                    }
                    H81[i + 9 * j] = Hq[i + 9 * j] / r - (gq[i] * gr[j] + gr[i] * gq[j]) / (r * r) + (2 * q / (r * r * r)) * gr[i] * gr[j]
                        - (q / (r * r)) * Hr;
                }
        const int coef[12] = { 0, 0, 0, 0, -1, 1, 0, 0, -1, 0, 1, 0 };
        expand(3, coef, G9, H81, g, H);
        return q / r;
    }
    double w[3], e[3], f[3], n[3], fxw[3], wxe[3];
    if (kind == K_PT) {
        sub3(X[0], X[1], w);
        sub3(X[2], X[1], e);
        sub3(X[3], X[1], f);
    }
    else {
        sub3(X[2], X[0], w);
        sub3(X[1], X[0], e);
        sub3(X[3], X[2], f);
    }
    cross3(e, f, n);
    const double s = dot3(w, n);
    q_derivs(e, f, &q, gq, Hq);
    cross3(f, w, fxw);
    cross3(w, e, wxe);
    const double gs[9] = { n[0], n[1], n[2], fxw[0], fxw[1], fxw[2], wxe[0], wxe[1], wxe[2] };
    const double c1 = 2 * s / q, c2 = s * s / (q * q);
    for (int i = 0; i < 9; ++i) G9[i] = c1 * gs[i] - c2 * gq[i];
    if (H) {
        double Hs[81];
        for (int i = 0; i < 81; ++i) Hs[i] = 0.0;
        add_skew(Hs, 0, 1, f, -1.0);
        add_skew(Hs, 1, 0, f, 1.0);
        add_skew(Hs, 0, 2, e, 1.0);
        add_skew(Hs, 2, 0, e, -1.0);
        add_skew(Hs, 1, 2, w, -1.0);
        add_skew(Hs, 2, 1, w, 1.0);
        for (int i = 0; i < 9; ++i)
            for (int j = 0; j < 9; ++j)
                H81[i + 9 * j] = (2 / q) * gs[i] * gs[j] + c1 * Hs[i + 9 * j] - (2 * s / (q * q)) * (gs[i] * gq[j] + gq[i] * gs[j])
                    + (2 * s * s / (q * q * q)) * gq[i] * gq[j] - c2 * Hq[i + 9 * j];
    }
    const int coefPT[12] = { 1, -1, 0, 0, 0, -1, 1, 0, 0, -1, 0, 1 };
    const int coefEE[12] = { -1, 0, 1, 0, -1, 1, 0, 0, 0, 0, -1, 1 };
    expand(4, kind == K_PT ? coefPT : coefEE, G9, H81, g, H);
    return s * s / q;
}

// c = |(v1-v0) x (v3-v2)|^2 with gradient / Hessian over the four nodes
__device__ inline double cross_sqnorm_derivs(const double (*X)[3], double* g, double* H)
{
    double e[3], f[3], q, gq[9], Hq[81];
    sub3(X[1], X[0], e);
    sub3(X[3], X[2], f);
    q_derivs(e, f, &q, gq, Hq);
    const int coef[12] = { 0, 0, 0, 0, -1, 1, 0, 0, 0, 0, -1, 1 };
    expand(4, coef, gq, Hq, g, H);
    return q;
}

} // namespace cdev
} // namespace ipcgpu
