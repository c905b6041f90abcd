// PSD-projected barrier Hessian of ONE contact stencil, formed and projected in the 3 (nn - 1)-dimensional complement of the rigid
// translations, every index a compile-time constant (gfx950: everything below lives in registers, no scratch).
//   SelfCollisionHandler::augmentIPHessian / augmentParaEEHessian   SelfCollisionHandler.cpp:418-561, 3039-3201
//   derivatives of the squared distances                            MeshCollisionUtils.hpp:163-2015 (there: MATLAB-generated g_* / H_*)
//   IglUtils::makePD                                                IglUtils.hpp:119-137
//
// Round 6.  Until round 5 a lane formed the 12 x 12 node-space block (stencil_distance -> expand, contact_device.h) with a RUN-TIME stencil size,
// reduced it to 9 x 9 for the Jacobi sweeps and expanded it again: two 144-double arrays per lane walked with run-time indices = 4 960 B of scratch
// per lane.  Here the stencil kind is a template parameter (the MMCVID sign encoding gives it for free; the host bins the set by kind), and the
// node-space block is never formed:
//
//   * every squared distance is a function of node DIFFERENCES y_u = sum_k c[u][k] x_k (point-point: r; point-edge: e, f; point-triangle and
//     edge-edge: w, e, f), and so is the cross norm of the mollifier;
//   * with the Helmert basis R = H (x) I3 of the complement of the translations (orthonormal columns), z = R^T x, the differences are y_u = sum_a T[u][a] z_a
//     with the 3 x 3 constant T = c H^T of the kind, and makePD(B) = R makePD(R^T B R) R^T exactly (B annihilates the translations);
//   * C = R^T B R is accumulated directly from a handful of reduced vectors: for d = s^2 / q (s = w . (e x f), q = |e x f|^2)
//         B_y = alpha gs gs^T + beta (gs gq^T + gq gs^T) + gamma gq gq^T + kappa m b' (c1 Hs - c2 Hq),
//     Hs = the skew blocks of the triple product (its reduced form is skew(m_ab), m_ab from the 2 x 2 minors of T), Hq = Hessian of |e x f|^2;
//     the point-edge form d = q / r is the same with r = |f - e|^2 in the role of s;
//   * the 45 + 81 scalars of the Jacobi iteration (jacobi9_device.h, now for M = 3, 6, 9) are the only big live set;
//   * the projected block leaves as the <= 10 node-pair blocks A_kl = sum_ab h[a][k] h[b][l] C+[a][b], written straight into the scatter slots.
//
// The mollified (parallel edge-edge) stencils use the same pieces with a RUN-TIME T for their distance part: which of the four edge nodes the
// distance stencil names arrives as 0 / 1 selection coefficients that multiply, never as indices.
//
// Compiles for the host too (tests/test_stencil_hessian.py: against the oracle's derivatives + numpy's eigh).
#pragma once
#include "jacobi9_device.h"

namespace ipcgpu {
namespace sh {

#define SH_HD J9_HD

enum { KIND_PP = 0, KIND_PE = 1, KIND_PT = 2, KIND_EE = 3 };

constexpr double S2 = 0.70710678118654752440, S6 = 0.40824829046386301637, S12 = 0.28867513459481288225;
// Helmert rows h[a][k]: orthonormal, orthogonal to (1, ..., 1); a stencil of nn nodes uses rows a < nn - 1 and columns k < nn
constexpr double H4[3][4] = { { S2, -S2, 0.0, 0.0 }, { S6, S6, -2.0 * S6, 0.0 }, { S12, S12, S12, -3.0 * S12 } };

// node differences of the kinds: y_u = sum_k COEF[kind][u][k] x_k
constexpr int COEF[4][3][4] = {
    { { 1, -1, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 } }, // PP: r = x0 - x1
    { { 0, 0, 0, 0 }, { -1, 1, 0, 0 }, { -1, 0, 1, 0 } }, // PE: e = x1 - x0, f = x2 - x0
    { { 1, -1, 0, 0 }, { 0, -1, 1, 0 }, { 0, -1, 0, 1 } }, // PT: w = x0 - x1, e = x2 - x1, f = x3 - x1
    { { -1, 0, 1, 0 }, { -1, 1, 0, 0 }, { 0, 0, -1, 1 } }, // EE: w = x2 - x0, e = x1 - x0, f = x3 - x2
};
constexpr int NN_OF[4] = { 2, 3, 4, 4 };

// T[u][a] = sum_k COEF[kind][u][k] H4[a][k]
constexpr double t_entry(int kind, int u, int a)
{
    double s = 0.0;
    for (int k = 0; k < 4; ++k) s += COEF[kind][u][k] * H4[a][k];
    return s;
}

SH_HD void cross(const double* a, const double* b, double* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
SH_HD double dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// symmetric matrices of dimension D = 3 NR: upper triangle of a D x D array (the other entries are never named)
template <int D>
constexpr int su(int i, int j) { return i <= j ? i * D + j : j * D + i; }

// C += s v v^T
template <int NR, int SD = 3 * NR>
SH_HD void add_rank1(double* C, double s, const double* v)
{
    constexpr int D = 3 * NR;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const double sv = s * v[i];
#pragma unroll
        for (int j = i; j < D; ++j) C[su<SD>(i, j)] += sv * v[j];
    }
}
// C += s (u v^T + v u^T)
template <int NR, int SD = 3 * NR>
SH_HD void add_rank2(double* C, double s, const double* u, const double* v)
{
    constexpr int D = 3 * NR;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const double su_ = s * u[i], sv = s * v[i];
#pragma unroll
        for (int j = i; j < D; ++j) C[su<SD>(i, j)] += su_ * v[j] + sv * u[j];
    }
}
// v_r[a] = sum_u t_u[a] v_u for two difference vectors (rows t1, t2 of T) -- the reduced form of a y-space vector with blocks (v1, v2)
template <int NR>
SH_HD void reduce2(const double* t1, const double* t2, const double* v1, const double* v2, double* vr)
{
#pragma unroll
    for (int a = 0; a < NR; ++a)
#pragma unroll
        for (int i = 0; i < 3; ++i) vr[3 * a + i] = t1[a] * v1[i] + t2[a] * v2[i];
}
// C += s T^T Hq T, Hq = Hessian of q(e, f) = |e x f|^2 in (e, f): blocks ee: 2 ff I - 2 f f^T, ff: 2 ee I - 2 e e^T, ef: 4 e f^T - 2 f e^T - 2 ef I
template <int NR, int SD = 3 * NR>
SH_HD void add_hq(double* C, double s, const double* t1, const double* t2, const double* e, const double* f)
{
    constexpr int D = 3 * NR;
    const double ee = dot(e, e), ff = dot(f, f), ef = dot(e, f);
#pragma unroll
    for (int a = 0; a < NR; ++a)
#pragma unroll
        for (int b = a; b < NR; ++b) {
            const double c11 = s * (t1[a] * t1[b]), c22 = s * (t2[a] * t2[b]), c12 = s * (t1[a] * t2[b]), c21 = s * (t2[a] * t1[b]);
            const double cI = 2.0 * (c11 * ff + c22 * ee - (c12 + c21) * ef);
            const double kff = -2.0 * c11, kee = -2.0 * c22, kef = 4.0 * c12 - 2.0 * c21, kfe = 4.0 * c21 - 2.0 * c12;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (a == b && j < i) continue;
                    double v = kff * f[i] * f[j] + kee * e[i] * e[j] + kef * e[i] * f[j] + kfe * f[i] * e[j];
                    if (i == j) v += cI;
                    C[su<SD>(3 * a + i, 3 * b + j)] += v;
                }
        }
}
// C += s T^T Hs T, Hs = Hessian of the triple product s = w . (e x f) in (w, e, f): blocks we: -[f]x, wf: [e]x, ef: -[w]x (and their transposes).
// Block (a, b) of the reduced form is the skew matrix of m_ab = -M01 f + M02 e - M12 w, M_uv = T[u][a] T[v][b] - T[v][a] T[u][b]; m_aa = 0.
SH_HD void add_hs(double* C, double s, const double* t0, const double* t1, const double* t2, const double* w, const double* e, const double* f)
{
    constexpr int SD = 9;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = a + 1; b < 3; ++b) {
            const double m01 = s * (t0[a] * t1[b] - t1[a] * t0[b]), m02 = s * (t0[a] * t2[b] - t2[a] * t0[b]), m12 = s * (t1[a] * t2[b] - t2[a] * t1[b]);
            double m[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) m[i] = m02 * e[i] - m01 * f[i] - m12 * w[i];
            // [m]x = [[0, -m2, m1], [m2, 0, -m0], [-m1, m0, 0]]
            C[su<SD>(3 * a + 0, 3 * b + 1)] -= m[2];
            C[su<SD>(3 * a + 0, 3 * b + 2)] += m[1];
            C[su<SD>(3 * a + 1, 3 * b + 0)] += m[2];
            C[su<SD>(3 * a + 1, 3 * b + 2)] -= m[0];
            C[su<SD>(3 * a + 2, 3 * b + 0)] -= m[1];
            C[su<SD>(3 * a + 2, 3 * b + 1)] += m[0];
        }
}

// ---- the three distance forms: C += cH G G^T + cG H (reduced), Gr = reduced gradient of d.  cH = (weight) b'', cG = (weight) b' -------------------
// point-point: d = |r|^2, y = r with coefficient row t
template <int NR, int SD = 3 * NR>
SH_HD void form_pp(double* C, double* Gr, const double* t, const double* r, double cH, double cG)
{
    constexpr int D = 3 * NR;
#pragma unroll
    for (int a = 0; a < NR; ++a)
#pragma unroll
        for (int i = 0; i < 3; ++i) Gr[3 * a + i] = 2.0 * t[a] * r[i];
    add_rank1<NR, SD>(C, cH, Gr);
#pragma unroll
    for (int a = 0; a < NR; ++a)
#pragma unroll
        for (int b = a; b < NR; ++b) {
            const double v = cG * 2.0 * (t[a] * t[b]);
#pragma unroll
            for (int i = 0; i < 3; ++i) C[su<SD>(3 * a + i, 3 * b + i)] += v;
        }
}
SH_HD double dist_pp(const double* r) { return dot(r, r); }
// point-edge: d = q / r, q = |e x f|^2, r = |f - e|^2
SH_HD double dist_pe(const double* e, const double* f)
{
    double n[3];
    cross(e, f, n);
    const double g[3] = { f[0] - e[0], f[1] - e[1], f[2] - e[2] };
    return dot(n, n) / dot(g, g);
}
template <int NR, int SD = 3 * NR>
SH_HD void form_pe(double* C, double* Gr, const double* t1, const double* t2, const double* e, const double* f, double cH, double cG)
{
    constexpr int D = 3 * NR;
    double n[3], fxn[3], nxe[3];
    cross(e, f, n);
    cross(f, n, fxn);
    cross(n, e, nxe);
    const double g[3] = { f[0] - e[0], f[1] - e[1], f[2] - e[2] };
    const double q = dot(n, n), r = dot(g, g), ir = 1.0 / r;
    double gq[3 * NR], gr[3 * NR], td[NR];
    const double gqe[3] = { 2.0 * fxn[0], 2.0 * fxn[1], 2.0 * fxn[2] }, gqf[3] = { 2.0 * nxe[0], 2.0 * nxe[1], 2.0 * nxe[2] };
    reduce2<NR>(t1, t2, gqe, gqf, gq);
#pragma unroll
    for (int a = 0; a < NR; ++a) {
        td[a] = t2[a] - t1[a];
#pragma unroll
        for (int i = 0; i < 3; ++i) gr[3 * a + i] = 2.0 * td[a] * g[i];
    }
    const double qr2 = q * ir * ir; // q / r^2
#pragma unroll
    for (int i = 0; i < D; ++i) Gr[i] = gq[i] * ir - qr2 * gr[i];
    add_rank1<NR, SD>(C, cH * ir * ir, gq);
    add_rank2<NR, SD>(C, -(cH * qr2 * ir) - cG * ir * ir, gq, gr);
    add_rank1<NR, SD>(C, cH * qr2 * qr2 + cG * 2.0 * qr2 * ir, gr);
    add_hq<NR, SD>(C, cG * ir, t1, t2, e, f);
    // -(q / r^2) Hr, Hr = 2 td td^T (x) I
#pragma unroll
    for (int a = 0; a < NR; ++a)
#pragma unroll
        for (int b = a; b < NR; ++b) {
            const double v = -(cG * qr2) * 2.0 * (td[a] * td[b]);
#pragma unroll
            for (int i = 0; i < 3; ++i) C[su<SD>(3 * a + i, 3 * b + i)] += v;
        }
}
// point-triangle / edge-edge: d = s^2 / q, s = w . (e x f), q = |e x f|^2
SH_HD double dist_tt(const double* w, const double* e, const double* f)
{
    double n[3];
    cross(e, f, n);
    const double s = dot(w, n);
    return s * s / dot(n, n);
}
SH_HD void form_tt(double* C, double* Gr, const double* t0, const double* t1, const double* t2, const double* w, const double* e, const double* f, double cH,
    double cG)
{
    double n[3], fxn[3], nxe[3], fxw[3], wxe[3];
    cross(e, f, n);
    cross(f, n, fxn);
    cross(n, e, nxe);
    cross(f, w, fxw);
    cross(w, e, wxe);
    const double s = dot(w, n), q = dot(n, n), iq = 1.0 / q;
    const double c1 = 2.0 * s * iq, c2 = s * s * iq * iq;
    double gs[9], gq[9];
    const double gqe[3] = { 2.0 * fxn[0], 2.0 * fxn[1], 2.0 * fxn[2] }, gqf[3] = { 2.0 * nxe[0], 2.0 * nxe[1], 2.0 * nxe[2] };
    reduce2<3>(t1, t2, gqe, gqf, gq);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int i = 0; i < 3; ++i) gs[3 * a + i] = t0[a] * n[i] + t1[a] * fxw[i] + t2[a] * wxe[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) Gr[i] = c1 * gs[i] - c2 * gq[i];
    add_rank1<3>(C, cH * c1 * c1 + cG * 2.0 * iq, gs);
    add_rank2<3>(C, -(cH * c1 * c2) - cG * 2.0 * s * iq * iq, gs, gq);
    add_rank1<3>(C, cH * c2 * c2 + cG * 2.0 * s * s * iq * iq * iq, gq);
    add_hs(C, cG * c1, t0, t1, t2, w, e, f);
    add_hq<3>(C, -(cG * c2), t1, t2, e, f);
}

SH_HD void barrier_derivs(double d, double dHat, double* b, double* gb, double* Hb)
{
    // BarrierFunctions.hpp:56-83 (the same expressions as cdev::barrier)
    const double t2 = d - dHat, lg = log(d / dHat);
    *b = -t2 * t2 * lg;
    *gb = t2 * lg * -2.0 - (t2 * t2) / d;
    *Hb = (lg * -2.0 - t2 * 4.0 / d) + 1.0 / (d * d) * (t2 * t2);
}

// ---- projection and the way back to node pairs ------------------------------------------------------------------------------------------
// C (sym upper, M = 3 NR) -> C+ = V max(lambda, 0) V^T in place; returns the number of Jacobi sweeps of this lane
template <int M>
SH_HD int project_psd(double (&C)[M * M])
{
    double V[M * M];
    const int sweeps = j9::jacobi_sweeps<M>(C, V);
    double ev[M];
#pragma unroll
    for (int i = 0; i < M; ++i) ev[i] = C[su<M>(i, i)] > 0.0 ? C[su<M>(i, i)] : 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j)
#pragma unroll
        for (int i = 0; i <= j; ++i) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < M; ++k) v += V[i + M * k] * ev[k] * V[j + M * k];
            C[su<M>(i, j)] = v;
        }
    return sweeps;
}

// node-pair block (k, l) of R C+ R^T: A[r + 3 c] = sum_ab h[a][k] h[b][l] C+[3 a + r, 3 b + c]   (NR = NN - 1 rows of the Helmert table)
template <int NR, int K, int L, int SD = 3 * NR>
SH_HD void pair_block(const double* C, double* A)
{
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double v = 0.0;
#pragma unroll
            for (int a = 0; a < NR; ++a)
#pragma unroll
                for (int b = 0; b < NR; ++b) {
                    constexpr double zero = 0.0;
                    const double hh = H4[a][K] * H4[b][L];
                    if (hh != zero) v += hh * C[su<SD>(3 * a + r, 3 * b + c)]; // (constant after unrolling: the zero columns of the Helmert rows cost nothing)
                }
            A[r + 3 * c] = v;
        }
}

// The reduced block of an ACTIVE stencil of kind KIND at node positions X (nn x 3): C (sym upper, 3 (nn - 1)) = kappa mult (b'' g g^T + b' Hess d),
// not yet projected; stored with leading dimension SD (SD = 9: every kind in the 9 x 9 frame of the Jacobi iteration, rows / columns >= 3 (nn - 1) zero).
// Returns d.
template <int KIND, int SD = 3 * (NN_OF[KIND] - 1)>
SH_HD double active_block(const double (*X)[3], double dHat, double weight, double* C)
{
    constexpr int NN = NN_OF[KIND], NR = NN - 1, D = 3 * NR;
#pragma unroll
    for (int i = 0; i < SD * SD; ++i) C[i] = 0.0;
    double y[3][3];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < NN; ++k)
                if (COEF[KIND][u][k] != 0) v += COEF[KIND][u][k] * X[k][c];
            y[u][c] = v;
        }
    double t[3][3];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int a = 0; a < 3; ++a) t[u][a] = (a < NR) ? t_entry(KIND, u, a) : 0.0;
    double b, gb, Hb, Gr[D];
    if constexpr (KIND == KIND_PP) {
        const double d = dist_pp(y[0]);
        barrier_derivs(d, dHat, &b, &gb, &Hb);
        form_pp<NR, SD>(C, Gr, t[0], y[0], weight * Hb, weight * gb);
        return d;
    }
    else if constexpr (KIND == KIND_PE) {
        const double d = dist_pe(y[1], y[2]);
        barrier_derivs(d, dHat, &b, &gb, &Hb);
        form_pe<NR, SD>(C, Gr, t[1], t[2], y[1], y[2], weight * Hb, weight * gb);
        return d;
    }
    else {
        const double d = dist_tt(y[0], y[1], y[2]);
        barrier_derivs(d, dHat, &b, &gb, &Hb);
        form_tt(C, Gr, t[0], t[1], t[2], y[0], y[1], y[2], weight * Hb, weight * gb);
        return d;
    }
}

// The reduced block (9 x 9, on the four nodes XE of the edge pair) of a MOLLIFIED stencil: kappa (b' e' (gd gc^T + gc gd^T) + b (e' Hc + e'' gc gc^T)
// + e b'' gd gd^T + e b' Hd)   (SelfCollisionHandler.cpp:3105-3169).  The distance stencil (kind, its nodes among the four) arrives as sel[k][q] = 1.0 where
// its node k IS edge node q, 0.0 elsewhere.  eps_x: the mollifier threshold of the pair (MeshCollisionUtils.hpp:2969-2974).
template <int KIND>
SH_HD void para_block(const double (*XE)[3], const double (*sel)[4], double dHat, double kappa, double eps_x, double* C)
{
#pragma unroll
    for (int i = 0; i < 81; ++i) C[i] = 0.0;
    // the cross norm c = |(x1 - x0) x (x3 - x2)|^2 lives in the edge-edge differences e, f
    double tE[3][3], yE[3][3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
#pragma unroll
        for (int a = 0; a < 3; ++a) tE[u][a] = t_entry(KIND_EE, u, a);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (COEF[KIND_EE][u][k] != 0) v += COEF[KIND_EE][u][k] * XE[k][c];
            yE[u][c] = v;
        }
    }
    double n[3], fxn[3], nxe[3], gc[9];
    cross(yE[1], yE[2], n);
    cross(yE[2], n, fxn);
    cross(n, yE[1], nxe);
    const double cn = dot(n, n);
    {
        const double gqe[3] = { 2.0 * fxn[0], 2.0 * fxn[1], 2.0 * fxn[2] }, gqf[3] = { 2.0 * nxe[0], 2.0 * nxe[1], 2.0 * nxe[2] };
        reduce2<3>(tE[1], tE[2], gqe, gqf, gc);
    }
    double em, eg, eH; // mollifier (MeshCollisionUtils.hpp:2834-2866; the same expressions as cdev::mollifier)
    if (cn < eps_x) {
        const double r = cn / eps_x;
        em = (-r + 2.0) * r;
        eg = 2.0 * (1.0 / eps_x) * (-(1.0 / eps_x) * cn + 1.0);
        eH = -2.0 / (eps_x * eps_x);
    }
    else {
        em = 1.0;
        eg = 0.0;
        eH = 0.0;
    }
    // the distance stencil in ITS differences, mapped onto the four edge nodes: crun[u][q] = sum_k COEF[kind][u][k] sel[k][q], T = crun H^T
    double t[3][3], y[3][3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        double crun[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (COEF[KIND][u][k] != 0) v += (double)COEF[KIND][u][k] * sel[k][q];
            crun[q] = v;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) t[u][a] = crun[0] * H4[a][0] + crun[1] * H4[a][1] + crun[2] * H4[a][2] + crun[3] * H4[a][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) y[u][c] = crun[0] * XE[0][c] + crun[1] * XE[1][c] + crun[2] * XE[2][c] + crun[3] * XE[3][c];
    }
    double b, gb, Hb, gd[9];
    if constexpr (KIND == KIND_PP) {
        barrier_derivs(dist_pp(y[0]), dHat, &b, &gb, &Hb);
        form_pp<3>(C, gd, t[0], y[0], kappa * em * Hb, kappa * em * gb);
    }
    else if constexpr (KIND == KIND_PE) {
        barrier_derivs(dist_pe(y[1], y[2]), dHat, &b, &gb, &Hb);
        form_pe<3>(C, gd, t[1], t[2], y[1], y[2], kappa * em * Hb, kappa * em * gb);
    }
    else {
        barrier_derivs(dist_tt(y[0], y[1], y[2]), dHat, &b, &gb, &Hb);
        form_tt(C, gd, t[0], t[1], t[2], y[0], y[1], y[2], kappa * em * Hb, kappa * em * gb);
    }
    add_rank2<3>(C, kappa * gb * eg, gd, gc);
    add_rank1<3>(C, kappa * b * eH, gc);
    add_hq<3>(C, kappa * b * eg, tE[1], tE[2], yE[1], yE[2]);
}

} // namespace sh
} // namespace ipcgpu
