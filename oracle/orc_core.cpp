// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_api.h).
// Mesh features, neo-Hookean elasticity, CSR pattern / assembly, inversion step bound.
#include "orc_api.h"
#include "orc_core.h"
#include <complex>
#include <cstdio>
#include <cassert>

namespace orc {

// ---------------------------------------------------------------- NH in sigma space
// NeoHookeanEnergy.cpp:55-69
static double nh_E(const double s[3], double u, double lam)
{
    if (u == 0.0 && lam == 0.0) return 0.0;
    const double sigma2Sum = s[0] * s[0] + s[1] * s[1] + s[2] * s[2];
    const double sigmaProd = s[0] * s[1] * s[2];
    const double L = std::log(sigmaProd);
    return u / 2.0 * (sigma2Sum - 3) - (u - lam / 2.0 * L) * L;
}
// NeoHookeanEnergy.cpp:71-90
static void nh_dE(const double s[3], double u, double lam, double dE[3])
{
    if (u == 0.0 && lam == 0.0) {
        dE[0] = dE[1] = dE[2] = 0;
        return;
    }
    const double L = std::log(s[0] * s[1] * s[2]);
    for (int i = 0; i < 3; ++i) {
        const double inv = 1.0 / s[i];
        dE[i] = u * (s[i] - inv) + lam * inv * L;
    }
}
// NeoHookeanEnergy.cpp:91-114
static void nh_d2E(const double s[3], double u, double lam, double d2[9])
{
    if (u == 0.0 && lam == 0.0) {
        for (int i = 0; i < 9; ++i) d2[i] = 0;
        return;
    }
    const double L = std::log(s[0] * s[1] * s[2]);
    for (int i = 0; i < 3; ++i) {
        const double inv2 = 1.0 / s[i] / s[i];
        d2[i + 3 * i] = u * (1.0 + inv2) - lam * inv2 * (L - 1.0);
    }
    d2[0 + 3 * 1] = d2[1 + 3 * 0] = lam / s[0] / s[1];
    d2[1 + 3 * 2] = d2[2 + 3 * 1] = lam / s[1] / s[2];
    d2[2 + 3 * 0] = d2[0 + 3 * 2] = lam / s[2] / s[0];
}
// NeoHookeanEnergy.cpp:115-136
static void nh_BLeft(const double s[3], double u, double lam, double B[3])
{
    if (u == 0.0 && lam == 0.0) {
        B[0] = B[1] = B[2] = 0;
        return;
    }
    const double sigmaProd = s[0] * s[1] * s[2];
    const double middle = u - lam * std::log(sigmaProd);
    B[0] = (u + middle / s[0] / s[1]) / 2.0;
    B[1] = (u + middle / s[1] / s[2]) / 2.0;
    B[2] = (u + middle / s[2] / s[0]) / 2.0;
}
// NeoHookeanEnergy.cpp:138-153: P from F directly, J from the singular values
static M3 nh_P(const M3& F, const double s[3], double u, double lam)
{
    M3 P;
    if (u == 0.0 && lam == 0.0) {
        for (int i = 0; i < 9; ++i) P.m[i] = 0;
        return P;
    }
    const double J = s[0] * s[1] * s[2];
    M3 FInvT = cofactor(F);
    for (int i = 0; i < 9; ++i) FInvT.m[i] /= J;
    const double lJ = std::log(J);
    for (int i = 0; i < 9; ++i) P.m[i] = u * (F.m[i] - FInvT.m[i]) + lam * lJ * FInvT.m[i];
    return P;
}

// ---------------------------------------------------------------- fixed corotated in sigma space
// FixedCoRotEnergy.cpp:62-70
static double fcr_E(const double s[3], double u, double lam)
{
    const double d0 = s[0] - 1.0, d1 = s[1] - 1.0, d2 = s[2] - 1.0;
    const double sigmam12Sum = d0 * d0 + d1 * d1 + d2 * d2;
    const double sigmaProdm1 = s[0] * s[1] * s[2] - 1.0;
    return u * sigmam12Sum + lam / 2.0 * sigmaProdm1 * sigmaProdm1;
}
// FixedCoRotEnergy.cpp:72-95
static void fcr_dE(const double s[3], double u, double lam, double dE[3])
{
    const double sigmaProdm1lambda = lam * (s[0] * s[1] * s[2] - 1.0);
    const double noI[3] = { s[1] * s[2], s[2] * s[0], s[0] * s[1] };
    const double _2u = u * 2;
    for (int i = 0; i < 3; ++i) dE[i] = _2u * (s[i] - 1.0) + noI[i] * sigmaProdm1lambda;
}
// FixedCoRotEnergy.cpp:96-128
static void fcr_d2E(const double s[3], double u, double lam, double d2[9])
{
    const double sigmaProd = s[0] * s[1] * s[2];
    const double noI[3] = { s[1] * s[2], s[2] * s[0], s[0] * s[1] };
    const double _2u = u * 2;
    for (int i = 0; i < 3; ++i) d2[i + 3 * i] = _2u + lam * noI[i] * noI[i];
    d2[0 + 3 * 1] = d2[1 + 3 * 0] = lam * (s[2] * (sigmaProd - 1.0) + noI[0] * noI[1]);
    d2[0 + 3 * 2] = d2[2 + 3 * 0] = lam * (s[1] * (sigmaProd - 1.0) + noI[0] * noI[2]);
    d2[2 + 3 * 1] = d2[1 + 3 * 2] = lam * (s[0] * (sigmaProd - 1.0) + noI[2] * noI[1]);
}
// FixedCoRotEnergy.cpp:129-144
static void fcr_BLeft(const double s[3], double u, double lam, double B[3])
{
    const double sigmaProd = s[0] * s[1] * s[2];
    const double halfLambda = lam / 2.0;
    B[0] = u - halfLambda * s[2] * (sigmaProd - 1);
    B[1] = u - halfLambda * s[0] * (sigmaProd - 1);
    B[2] = u - halfLambda * s[1] * (sigmaProd - 1);
}
// FixedCoRotEnergy.cpp:145-153: P = 2u (F - U V^T) + lam (prod sigma - 1) cof F
static M3 fcr_P(const M3& F, const M3& U, const double s[3], const M3& V, double u, double lam)
{
    M3 R; // U V^T
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R(i, j) = U(i, 0) * V(j, 0) + U(i, 1) * V(j, 1) + U(i, 2) * V(j, 2);
    M3 C = cofactor(F);
    const double k = lam * (s[0] * s[1] * s[2] - 1);
    M3 P;
    for (int i = 0; i < 9; ++i) P.m[i] = u * 2 * (F.m[i] - R.m[i]) + k * C.m[i];
    return P;
}

// dispatch on Config energyType ("energy NH" / "energy FCR", Config.cpp:23-24,107-111)
static double sig_E(int type, const double s[3], double u, double lam) { return type == 1 ? fcr_E(s, u, lam) : nh_E(s, u, lam); }
static void sig_dE(int type, const double s[3], double u, double lam, double dE[3]) { type == 1 ? fcr_dE(s, u, lam, dE) : nh_dE(s, u, lam, dE); }
static void sig_d2E(int type, const double s[3], double u, double lam, double d2[9]) { type == 1 ? fcr_d2E(s, u, lam, d2) : nh_d2E(s, u, lam, d2); }
static void sig_BLeft(int type, const double s[3], double u, double lam, double B[3]) { type == 1 ? fcr_BLeft(s, u, lam, B) : nh_BLeft(s, u, lam, B); }

// Energy.cpp:448-562. dPdF is 9x9 column-major with row-major vec index 3*i+j for F(i,j).
static void nh_dPdF(const M3& U, const double s[3], const M3& V, double u, double lam,
    double w, bool projectSPD, double dPdF[81], int type = 0)
{
    double dE[3], d2[9], BL[3];
    sig_dE(type, s, u, lam, dE);
    sig_d2E(type, s, u, lam, d2);
    if (projectSPD) make_pd(3, d2);
    sig_BLeft(type, s, u, lam, BL);
    double B[3][4];
    for (int cI = 0; cI < 3; ++cI) {
        int cP = (cI + 1) % 3;
        double rightCoef = dE[cI] + dE[cP];
        double sum_sigma = s[cI] + s[cP];
        const double eps = 1.0e-6;
        if (sum_sigma < eps) rightCoef /= 2.0 * eps;
        else rightCoef /= 2.0 * sum_sigma;
        const double leftCoef = BL[cI];
        B[cI][0] = B[cI][3] = leftCoef + rightCoef;
        B[cI][1] = B[cI][2] = leftCoef - rightCoef;
        if (projectSPD) make_pd2d(B[cI]);
    }
    double M[81];
    for (int i = 0; i < 81; ++i) M[i] = 0;
    auto Mx = [&](int i, int j) -> double& { return M[i + 9 * j]; };
    auto Bx = [&](int k, int i, int j) { return B[k][i + 2 * j]; };
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Mx(4 * i, 4 * j) = w * d2[i + 3 * j];
    Mx(1, 1) = w * Bx(0, 0, 0); Mx(1, 3) = w * Bx(0, 0, 1); Mx(3, 1) = w * Bx(0, 1, 0); Mx(3, 3) = w * Bx(0, 1, 1);
    Mx(5, 5) = w * Bx(1, 0, 0); Mx(5, 7) = w * Bx(1, 0, 1); Mx(7, 5) = w * Bx(1, 1, 0); Mx(7, 7) = w * Bx(1, 1, 1);
    Mx(2, 2) = w * Bx(2, 1, 1); Mx(2, 6) = w * Bx(2, 1, 0); Mx(6, 2) = w * Bx(2, 0, 1); Mx(6, 6) = w * Bx(2, 0, 0);
    // the 21 structurally non-zero entries of M (Energy.cpp:552), (row=3a+b, col=3c+d)
    static const int nzr[21] = { 0, 0, 0, 4, 4, 4, 8, 8, 8, 1, 1, 3, 3, 5, 5, 7, 7, 2, 2, 6, 6 };
    static const int nzc[21] = { 0, 4, 8, 0, 4, 8, 0, 4, 8, 1, 3, 1, 3, 5, 7, 5, 7, 2, 6, 2, 6 };
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            int ij = 3 * i + j;
            for (int r = 0; r < 3; ++r)
                for (int q = 0; q < 3; ++q) {
                    int rs = 3 * r + q;
                    if (ij > rs) continue;
                    double acc = 0;
                    for (int t = 0; t < 21; ++t) {
                        int a = nzr[t] / 3, b = nzr[t] % 3, c = nzc[t] / 3, d = nzc[t] % 3;
                        acc += Mx(nzr[t], nzc[t]) * U(i, a) * V(j, b) * U(r, c) * V(q, d);
                    }
                    dPdF[ij + 9 * rs] = acc;
                    if (ij < rs) dPdF[rs + 9 * ij] = acc;
                }
        }
}

// IglUtils.hpp:417-430 (DIM==3 branch): result(12 x cols) from right(9 x cols), row index 3*i+j
static void dF_div_dx_mult(int cols, const double* right /*9 x cols col-major*/, const M3& A, double* result /*12 x cols*/)
{
    for (int c = 0; c < cols; ++c) {
        const double* r = right + 9 * c;
        double* o = result + 12 * c;
        for (int k = 0; k < 3; ++k) // node k+1
            for (int i = 0; i < 3; ++i) { // component i
                o[3 + 3 * k + i] = A(k, 0) * r[3 * i + 0] + A(k, 1) * r[3 * i + 1] + A(k, 2) * r[3 * i + 2];
            }
        o[0] = -o[3] - o[6] - o[9];
        o[1] = -o[4] - o[7] - o[10];
        o[2] = -o[5] - o[8] - o[11];
    }
}

// ---------------------------------------------------------------- Mesh
Mesh::Mesh(int nV_, int nT_, const double* Vr, const int* Fc, double YM, double PR, double density_)
    : nV(nV_), nT(nT_), density(density_)
{
    V_rest.assign(Vr, Vr + 3 * nV);
    V = V_rest;
    F.assign(Fc, Fc + 4 * nT);
    dbcType.assign(nV, 0);
    restTriInv.resize(nT);
    triArea.resize(nT);
    vFLoc.assign(nV, {});
    vNeighbor.assign(nV, {});
    mass.assign(nV, 0.0);
    double edgeSum = 0;
    for (int t = 0; t < nT; ++t) {
        int v[4] = { Fi(t, 0), Fi(t, 1), Fi(t, 2), Fi(t, 3) };
        for (int k = 0; k < 4; ++k) vFLoc[v[k]].insert({ t, k });
        M3 X0;
        for (int k = 0; k < 3; ++k)
            for (int i = 0; i < 3; ++i) X0(i, k) = Vr[v[k + 1] + nV * i] - Vr[v[0] + nV * i];
        restTriInv[t] = inverse(X0);
        triArea[t] = det(X0) / 3 / 2; // Mesh.cpp:455
        // barycentric lumped mass (Mesh.cpp:255-266): |vol|/4 to each vertex
        double vol = std::fabs(det(X0)) / 6.0;
        for (int k = 0; k < 4; ++k) mass[v[k]] += vol / 4.0;
        for (int a = 0; a < 4; ++a)
            for (int b = a + 1; b < 4; ++b) {
                vNeighbor[v[a]].insert(v[b]);
                vNeighbor[v[b]].insert(v[a]);
            }
        for (int a = 0; a < 4; ++a) { // igl::avg_edge_length walks the columns cyclically: edges (0,1) (1,2) (2,3) (3,0)
            const int b = (a + 1) % 4;
            double d2 = 0;
            for (int i = 0; i < 3; ++i) {
                double d = Vr[v[a] + nV * i] - Vr[v[b] + nV * i];
                d2 += d * d;
            }
            edgeSum += std::sqrt(d2);
        }
    }
    // Mesh.cpp:460 igl::avg_edge_length(V_rest, F): the mean of |V(F(i,j)) - V(F(i,(j+1)%4))| over all i, j -- four of the six
    // edges of every tetrahedron.  The reference's spatial hash takes a third of it as its cell size, and the full CCD caps the
    // step with it (SpatialHash.hpp:603-618), so the value itself matters.
    avgEdgeLen = nT ? edgeSum / (4.0 * nT) : 0;
    for (int v = 0; v < nV; ++v) mass[v] *= density; // Mesh.cpp:399
    // Mesh.cpp:663-664
    mu.assign(nT, YM / 2.0 / (1.0 + PR));
    lam.assign(nT, YM * PR / (1.0 + PR) / (1.0 - 2.0 * PR));
    // bounding box of the simulated material (Mesh::matSpaceBBoxSize2, Mesh.cpp): nodes of elements only, so that a kinematic
    // obstacle riding along as a surface-only component does not change dHat = dHatEps^2 * diagonal^2
    inMesh.assign(nV, 0);
    for (int t = 0; t < nT; ++t)
        for (int k = 0; k < 4; ++k) inMesh[Fi(t, k)] = 1;
    meshBBox();
}

void Mesh::meshBBox()
{
    nElemNodes = 0;
    for (int v = 0; v < nV; ++v) nElemNodes += inMesh[v];
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    for (int v = 0; v < nV; ++v)
        for (int i = 0; i < 3; ++i) {
            if (nElemNodes && !inMesh[v]) continue;
            lo[i] = std::min(lo[i], V_rest[v + nV * i]);
            hi[i] = std::max(hi[i], V_rest[v + nV * i]);
        }
    bboxDiag2 = 0;
    for (int i = 0; i < 3; ++i) {
        bboxDiag2 += (hi[i] - lo[i]) * (hi[i] - lo[i]);
        bboxLo[i] = lo[i];
        bboxHi[i] = hi[i];
    }
}

void Mesh::setSurface(int n, const int* SFc, int nCE, const int* CE)
{
    nSF = n;
    SF.assign(SFc, SFc + 3 * n);
    std::set<std::pair<int, int>> es;
    std::set<int> svi;
    for (int f = 0; f < n; ++f) {
        int t[3] = { SFc[f], SFc[f + n], SFc[f + 2 * n] };
        for (int a = 0; a < 3; ++a) {
            svi.insert(t[a]);
            for (int b = a + 1; b < 3; ++b) {
                vNeighbor[t[a]].insert(t[b]);
                vNeighbor[t[b]].insert(t[a]);
            }
        }
        // Mesh.cpp:495-511
        if (!es.count({ t[1], t[0] })) es.insert({ t[0], t[1] });
        if (!es.count({ t[2], t[1] })) es.insert({ t[1], t[2] });
        if (!es.count({ t[0], t[2] })) es.insert({ t[2], t[0] });
    }
    SFEdges.assign(es.begin(), es.end());
    // codimensional segments (`.seg` shapes): vNeighbor (Mesh.cpp:490-493), appended to SFEdges behind the triangles' edges in file order
    // (:513-515), their ends on the surface (:912-915)
    for (int e = 0; e < nCE; ++e) {
        const int a = CE[2 * e], b = CE[2 * e + 1];
        vNeighbor[a].insert(b);
        vNeighbor[b].insert(a);
        SFEdges.emplace_back(a, b);
        svi.insert(a);
        svi.insert(b);
    }
    // nodes without any neighbour (`.pt` shapes) are surface vertices too (:916-920)
    codimPoints.clear();
    for (int v = 0; v < nV; ++v)
        if (vNeighbor[v].empty()) {
            svi.insert(v);
            codimPoints.push_back(v);
        }
    SVI.assign(svi.begin(), svi.end());
}

bool Mesh::checkInversion() const
{
    for (int t = 0; t < nT; ++t) {
        if (!(mu[t] && lam[t])) continue;
        M3 e;
        for (int k = 0; k < 3; ++k)
            for (int i = 0; i < 3; ++i) e(i, k) = Vx(Fi(t, k + 1), i) - Vx(Fi(t, 0), i);
        if (det(e) < 0.0) return false;
    }
    return true;
}

M3 Mesh::defGrad(int t) const
{
    M3 Xt;
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < 3; ++i) Xt(i, k) = Vx(Fi(t, k + 1), i) - Vx(Fi(t, 0), i);
    return mul(Xt, restTriInv[t]); // Energy.cpp:209-215
}

// Energy.cpp:195-242
double elasticEnergy(const Mesh& m, double coef, double* perElem)
{
    std::vector<double> e(m.nT);
#pragma omp parallel for schedule(static)
    for (int t = 0; t < m.nT; ++t) {
        M3 F = m.defGrad(t), U, V;
        double s[3];
        svd3(F, U, s, V);
        e[t] = sig_E(m.energyType, s, m.mu[t], m.lam[t]) * m.triArea[t];
    }
    double sum = 0;
    for (int t = 0; t < m.nT; ++t) sum += e[t]; // Eigen .sum() of the per-element vector (Energy.cpp:241)
    if (perElem) std::memcpy(perElem, e.data(), sizeof(double) * m.nT);
    return coef * sum;
}

// Energy.cpp:334-366
static void elemGradient(const Mesh& m, int t, double coef, double g[12])
{
    M3 F = m.defGrad(t), U, V;
    double s[3];
    svd3(F, U, s, V);
    M3 P = m.energyType == 1 ? fcr_P(F, U, s, V, m.mu[t], m.lam[t]) : nh_P(F, s, m.mu[t], m.lam[t]);
    const double w = coef * m.triArea[t];
    for (int i = 0; i < 9; ++i) P.m[i] *= w;
    const M3& A = m.restTriInv[t];
    // IglUtils.cpp:656-667: result[3+3k+i] = A.row(k) . P.row(i)
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < 3; ++i) g[3 + 3 * k + i] = A(k, 0) * P(i, 0) + A(k, 1) * P(i, 1) + A(k, 2) * P(i, 2);
    g[0] = -g[3] - g[6] - g[9];
    g[1] = -g[4] - g[7] - g[10];
    g[2] = -g[5] - g[8] - g[11];
}

// Energy.cpp:245-289
void elasticGradient(const Mesh& m, double coef, bool projectDBC, double* grad)
{
    std::vector<double> gc(12 * (size_t)m.nT);
#pragma omp parallel for schedule(static)
    for (int t = 0; t < m.nT; ++t) elemGradient(m, t, coef, &gc[12 * (size_t)t]);
#pragma omp parallel for schedule(static)
    for (int v = 0; v < m.nV; ++v) {
        double a[3] = { 0, 0, 0 };
        for (const auto& fl : m.vFLoc[v])
            for (int i = 0; i < 3; ++i) a[i] += gc[12 * (size_t)fl.first + 3 * fl.second + i];
        for (int i = 0; i < 3; ++i) grad[3 * v + i] = a[i];
    }
    if (projectDBC)
        for (int v = 0; v < m.nV; ++v)
            if (m.dbcType[v] != 0)
                for (int i = 0; i < 3; ++i) grad[3 * v + i] = 0; // every DBC vertex (Energy.cpp:284-288)
}

// Energy.cpp:368-408
void elemHessian(const Mesh& m, int t, double coef, bool projectSPD, double H[144])
{
    M3 F = m.defGrad(t), U, V;
    double s[3];
    svd3(F, U, s, V);
    const double w = coef * m.triArea[t];
    double dPdF[81];
    nh_dPdF(U, s, V, m.mu[t], m.lam[t], w, projectSPD, dPdF, m.energyType);
    const M3& A = m.restTriInv[t];
    double dPdF_T[81], wdPdx[12 * 9], wdPdx_T[9 * 12];
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) dPdF_T[i + 9 * j] = dPdF[j + 9 * i];
    dF_div_dx_mult(9, dPdF_T, A, wdPdx); // 12 x 9
    for (int i = 0; i < 12; ++i)
        for (int j = 0; j < 9; ++j) wdPdx_T[j + 9 * i] = wdPdx[i + 12 * j]; // 9 x 12
    dF_div_dx_mult(12, wdPdx_T, A, H); // 12 x 12
}

// ---------------------------------------------------------------- CSR pattern (LinSysSolver.hpp:46-150)
void Mesh::buildPattern()
{
    std::vector<std::set<int>> nb = vNeighbor;
    for (const auto& e : extraEdges) {
        nb[e.first].insert(e.second);
        nb[e.second].insert(e.first);
    }
    const int numRows = 3 * nV;
    ia.assign(numRows + 1, 0);
    std::vector<int> rowNNZ(numRows);
    for (int v = 0; v < nV; ++v) {
        int nnz = 3;
        for (int n : nb[v])
            if (n > v) nnz += 3;
        rowNNZ[3 * v] = nnz;
        rowNNZ[3 * v + 1] = nnz - 1;
        rowNNZ[3 * v + 2] = nnz - 2;
    }
    for (int r = 0; r < numRows; ++r) ia[r + 1] = ia[r] + rowNNZ[r];
    ja.assign(ia[numRows], 0);
    for (int v = 0; v < nV; ++v) {
        std::vector<int> cols;
        cols.push_back(3 * v);
        cols.push_back(3 * v + 1);
        cols.push_back(3 * v + 2);
        for (int n : nb[v])
            if (n > v) {
                cols.push_back(3 * n);
                cols.push_back(3 * n + 1);
                cols.push_back(3 * n + 2);
            }
        for (int r = 0; r < 3; ++r) {
            int p = ia[3 * v + r];
            for (size_t c = r; c < cols.size(); ++c) ja[p++] = cols[c];
        }
    }
}

int Mesh::findEntry(int row, int col) const
{
    const int* b = ja.data() + ia[row];
    const int* e = ja.data() + ia[row + 1];
    const int* it = std::lower_bound(b, e, col);
    if (it == e || *it != col) return -1;
    return int(it - ja.data());
}

// IglUtils.hpp:39-116 + LinSysSolver.hpp:331-339,402-410 (addCoeff silently ignores row > col)
static inline void addCoeff(const Mesh& m, double* a, int r, int c, double v)
{
    if (r <= c) {
        int k = m.findEntry(r, c);
        assert(k >= 0);
        a[k] += v;
    }
}
static inline void setCoeff(const Mesh& m, double* a, int r, int c, double v)
{
    if (r <= c) {
        int k = m.findEntry(r, c);
        assert(k >= 0);
        a[k] = v;
    }
}

void addBlockToMatrix(const Mesh& m, double* a, const double* H12 /*12x12 col-major*/, const int vInd[4], int rowIndI)
{
    int rowStart = vInd[rowIndI] * 3;
    if (rowStart < 0) {
        rowStart = -rowStart - 3;
        for (int i = 0; i < 3; ++i) setCoeff(m, a, rowStart + i, rowStart + i, 1.0);
        return;
    }
    for (int k = 0; k < 4; ++k) {
        if (vInd[k] < 0) continue;
        int c0 = vInd[k] * 3;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                addCoeff(m, a, rowStart + i, c0 + j, H12[(3 * rowIndI + i) + 12 * (3 * k + j)]);
    }
}

// Optimizer.cpp:3549-3668 restricted to elasticity + inertia (no contact, no damping)
void assembleHessian(const Mesh& m, double coef, bool projectDBC, double* a)
{
    const size_t nnz = m.ja.size();
    for (size_t i = 0; i < nnz; ++i) a[i] = 0; // setZero (3616)
    std::vector<double> Hs(144 * (size_t)m.nT);
    std::vector<int> vInds(4 * (size_t)m.nT);
#pragma omp parallel for schedule(static)
    for (int t = 0; t < m.nT; ++t) {
        elemHessian(m, t, coef, true, &Hs[144 * (size_t)t]);
        for (int k = 0; k < 4; ++k) {
            int v = m.Fi(t, k);
            vInds[4 * (size_t)t + k] = m.isProjectDBC(v, projectDBC) ? (-v - 1) : v; // Energy.cpp:402-407
        }
    }
#pragma omp parallel for schedule(static)
    for (int v = 0; v < m.nV; ++v)
        for (const auto& fl : m.vFLoc[v])
            addBlockToMatrix(m, a, &Hs[144 * (size_t)fl.first], &vInds[4 * (size_t)fl.first], fl.second);
    // Optimizer.cpp:3638-3668
#pragma omp parallel for schedule(static)
    for (int v = 0; v < m.nV; ++v) {
        if (!m.isProjectDBC(v, projectDBC)) {
            for (int i = 0; i < 3; ++i) addCoeff(m, a, 3 * v + i, 3 * v + i, m.mass[v]);
        }
        else {
            for (int i = 0; i < 3; ++i) setCoeff(m, a, 3 * v + i, 3 * v + i, 1.0);
        }
    }
}

// ---------------------------------------------------------------- inversion step bound (get_feasible_steps.cpp:9-172)
static double smallestPosRealQuadRoot(double a, double b, double c, double tol)
{
    double t;
    if (std::fabs(a) <= tol) t = -c / b;
    else {
        double desc = b * b - 4 * a * c;
        if (desc > 0) {
            t = (-b - std::sqrt(desc)) / (2 * a);
            if (t < 0) t = (-b + std::sqrt(desc)) / (2 * a);
        }
        else t = -1;
    }
    return t;
}
static double smallestPosRealCubicRoot(double a, double b, double c, double d, double tol)
{
    double t = -1;
    if (std::fabs(a) <= tol) t = smallestPosRealQuadRoot(b, c, d, tol);
    else {
        typedef std::complex<double> cd;
        cd i(0, 1);
        cd delta0(b * b - 3 * a * c, 0);
        cd delta1(2 * b * b * b - 9 * a * b * c + 27 * a * a * d, 0);
        cd C = std::pow((delta1 + std::sqrt(delta1 * delta1 - 4.0 * delta0 * delta0 * delta0)) / 2.0, 1.0 / 3.0);
        if (std::abs(C) == 0.0)
            C = std::pow((delta1 - std::sqrt(delta1 * delta1 - 4.0 * delta0 * delta0 * delta0)) / 2.0, 1.0 / 3.0);
        cd u2 = (-1.0 + std::sqrt(3.0) * i) / 2.0;
        cd u3 = (-1.0 - std::sqrt(3.0) * i) / 2.0;
        cd t1 = (b + C + delta0 / C) / (-3.0 * a);
        cd t2 = (b + u2 * C + delta0 / (u2 * C)) / (-3.0 * a);
        cd t3 = (b + u3 * C + delta0 / (u3 * C)) / (-3.0 * a);
        if ((std::fabs(std::imag(t1)) < tol) && (std::real(t1) > 0)) t = std::real(t1);
        if ((std::fabs(std::imag(t2)) < tol) && (std::real(t2) > 0) && ((std::real(t2) < t) || (t < 0))) t = std::real(t2);
        if ((std::fabs(std::imag(t3)) < tol) && (std::real(t3) > 0) && ((std::real(t3) < t) || (t < 0))) t = std::real(t3);
    }
    return t;
}

// get_feasible_steps.cpp:114-172 written through the compact determinant expansion of :175-209:
// det[x1-x0+t(p1-p0), ...] = a t^3 + b t^2 + c t + d0 ; solve a t^3 + b t^2 + c t + (1-slackness) d0 = 0.
void inversionStep(const Mesh& m, const double* p, double slackness, double* out)
{
    const double tol = 1.0e-6;
    for (int t = 0; t < m.nT; ++t) {
        double v[3][3], q[3][3];
        int i0 = m.Fi(t, 0);
        for (int k = 0; k < 3; ++k) {
            int ik = m.Fi(t, k + 1);
            for (int c = 0; c < 3; ++c) {
                v[k][c] = m.Vx(ik, c) - m.Vx(i0, c);
                q[k][c] = p[3 * ik + c] - p[3 * i0 + c];
            }
        }
        auto cross = [](const double* x, const double* y, double* z) {
            z[0] = x[1] * y[2] - x[2] * y[1];
            z[1] = x[2] * y[0] - x[0] * y[2];
            z[2] = x[0] * y[1] - x[1] * y[0];
        };
        auto dot = [](const double* x, const double* y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; };
        double pxp[3], vxp[3], pxv[3], vxv[3], mix[3];
        cross(q[0], q[1], pxp);
        cross(v[0], q[1], vxp);
        cross(q[0], v[1], pxv);
        cross(v[0], v[1], vxv);
        for (int c = 0; c < 3; ++c) mix[c] = vxp[c] + pxv[c];
        double a = dot(q[2], pxp);
        double b = dot(v[2], pxp) + dot(q[2], mix);
        double c = dot(q[2], vxv) + dot(v[2], mix);
        double d = (1.0 - slackness) * dot(v[2], vxv);
        double r = smallestPosRealCubicRoot(a, b, c, d, tol);
        out[t] = (r >= 0) ? r : 1e20;
    }
}

// Energy.cpp:565-581
double filterStepSize(const Mesh& m, const double* p, double stepSize)
{
    if (m.energyType == 1) return stepSize; // Energy.cpp:567: only energies that need the element-inversion safeguard
    std::vector<double> out(m.nT);
    inversionStep(m, p, 0.2, out.data());
    double mn = 1e300;
    for (double v : out) mn = std::min(mn, v);
    if (mn > 0.0 && mn < stepSize) stepSize = mn;
    return stepSize;
}

} // namespace orc

// ======================================================================= C API
using namespace orc;

extern "C" {

void orc_svd3(const double* F9, double* U9, double* S3, double* V9)
{
    M3 F, U, V;
    std::memcpy(F.m, F9, 72);
    svd3(F, U, S3, V);
    std::memcpy(U9, U.m, 72);
    std::memcpy(V9, V.m, 72);
}
void orc_make_pd(int n, double* A) { make_pd(n, A); }
void orc_nh_energy_sigma(const double* s3, double mu, double lam, double* E) { *E = nh_E(s3, mu, lam); }
void orc_nh_dPdF(const double* F9, double mu, double lam, double w, int projectSPD, double* out)
{
    M3 F, U, V;
    double s[3];
    std::memcpy(F.m, F9, 72);
    svd3(F, U, s, V);
    nh_dPdF(U, s, V, mu, lam, w, projectSPD != 0, out);
}
void orc_nh_P(const double* F9, double mu, double lam, double* P9)
{
    M3 F, U, V;
    double s[3];
    std::memcpy(F.m, F9, 72);
    svd3(F, U, s, V);
    M3 P = nh_P(F, s, mu, lam);
    std::memcpy(P9, P.m, 72);
}

orc_mesh* orc_mesh_create(int nV, int nT, const double* V, const int* F, double YM, double PR, double rho)
{
    return new orc_mesh(nV, nT, V, F, YM, PR, rho);
}
void orc_mesh_destroy(orc_mesh* h) { delete h; }
void orc_mesh_set_surface(orc_mesh* h, int nSF, const int* SF) { h->m.setSurface(nSF, SF); }
void orc_mesh_set_surface_codim(orc_mesh* h, int nSF, const int* SF, int nCE, const int* CE) { h->m.setSurface(nSF, SF, nCE, CE); }
void orc_mesh_set_dbc(orc_mesh* h, int n, const int* vids, int type)
{
    for (int i = 0; i < n; ++i) h->m.dbcType[vids[i]] = type;
}
// kinematic obstacle nodes (MeshCO riding along as a surface-only component); obstacleOnly: contact only with obstacles
void orc_mesh_set_obstacle(orc_mesh* h, int n, const int* vids, int obstacleOnly)
{
    Mesh& m = h->m;
    m.obstacle.assign(m.nV, 0);
    for (int i = 0; i < n; ++i) m.obstacle[vids[i]] = 1;
    m.obstacleOnly = obstacleOnly != 0;
}
void orc_mesh_set_codim_nodes(orc_mesh* h, int n, const int* vids, const double* nodeMass)
{
    // triangle meshes listed under `shapes` (componentCoDim 2): nodes of Mesh<3> with lumped masses = density x a third of the
    // adjacent triangle areas (Mesh.cpp:310-345, 399).  The Optimizer sizes dHat, eps_v, the Newton tolerance and kappa with
    // matSpaceBBoxSize2(dim) and avgNodeMass(dim), which run over the components of codimension 3 only (Mesh.cpp:576-637,
    // Optimizer.cpp:101, 2220, 2232) -- seen in a run of the reference-compiled code (2cubesFall_rotateCO_closedSurface.txt:
    // dHat = 1e-6 x 3, the box of the one tetrahedral cube).  So these nodes stay out of the box and the mean.
    Mesh& m = h->m;
    for (int i = 0; i < n; ++i) m.mass[vids[i]] = nodeMass[i];
    m.meshBBox();
}
void orc_mesh_set_component_material(orc_mesh* h, int nodeBegin, int nodeEnd, int tetBegin, int tetEnd, double rho, double YM, double PR)
{
    // Mesh::setLameParam with a componentMaterial entry (Mesh.cpp:661-671): nodal mass rescaled, Lame parameters of the element range
    Mesh& m = h->m;
    for (int v = nodeBegin; v < nodeEnd; ++v) m.mass[v] *= rho / m.density;
    for (int t = tetBegin; t < tetEnd; ++t) {
        m.mu[t] = YM / 2.0 / (1.0 + PR);
        m.lam[t] = YM * PR / (1.0 + PR) / (1.0 - 2.0 * PR);
    }
}
void orc_mesh_set_energy_type(orc_mesh* h, int type) { h->m.energyType = type; }
void orc_mesh_clear_dbc(orc_mesh* h) { std::fill(h->m.dbcType.begin(), h->m.dbcType.end(), 0); }
void orc_mesh_set_V(orc_mesh* h, const double* V) { h->m.V.assign(V, V + 3 * h->m.nV); }
void orc_mesh_get_V(const orc_mesh* h, double* V) { std::memcpy(V, h->m.V.data(), sizeof(double) * 3 * h->m.nV); }
void orc_mesh_get_features(const orc_mesh* h, double* A, double* vol, double* mass, double* mu, double* lam)
{
    const Mesh& m = h->m;
    if (A) for (int t = 0; t < m.nT; ++t) std::memcpy(A + 9 * t, m.restTriInv[t].m, 72);
    if (vol) std::memcpy(vol, m.triArea.data(), 8 * m.nT);
    if (mass) std::memcpy(mass, m.mass.data(), 8 * m.nV);
    if (mu) std::memcpy(mu, m.mu.data(), 8 * m.nT);
    if (lam) std::memcpy(lam, m.lam.data(), 8 * m.nT);
}
double orc_mesh_avg_edge_len(const orc_mesh* h) { return h->m.avgEdgeLen; }
double orc_mesh_bbox_diag2(const orc_mesh* h) { return h->m.bboxDiag2; }
int orc_mesh_check_inversion(const orc_mesh* h) { return h->m.checkInversion() ? 1 : 0; }

void orc_elastic_energy(const orc_mesh* h, double coef, double* E, double* perElem) { *E = elasticEnergy(h->m, coef, perElem); }
void orc_elastic_gradient(const orc_mesh* h, double coef, int projectDBC, double* g) { elasticGradient(h->m, coef, projectDBC != 0, g); }
void orc_elastic_hessian_elem(const orc_mesh* h, int e, double coef, int projectSPD, double* H) { elemHessian(h->m, e, coef, projectSPD != 0, H); }

int orc_pattern_build(orc_mesh* h)
{
    h->m.buildPattern();
    return (int)h->m.ja.size();
}
void orc_pattern_add_edges(orc_mesh* h, int n, const int* pairs)
{
    for (int i = 0; i < n; ++i) h->m.extraEdges.push_back({ pairs[2 * i], pairs[2 * i + 1] });
}
int orc_pattern_rows(const orc_mesh* h) { return 3 * h->m.nV; }
const int* orc_pattern_ia(const orc_mesh* h) { return h->m.ia.data(); }
const int* orc_pattern_ja(const orc_mesh* h) { return h->m.ja.data(); }
void orc_assemble_hessian(const orc_mesh* h, double coef, int projectDBC, double* a) { assembleHessian(h->m, coef, projectDBC != 0, a); }
void orc_csr_symv(const orc_mesh* h, const double* a, const double* x, double* y)
{
    const Mesh& m = h->m;
    int n = 3 * m.nV;
    for (int i = 0; i < n; ++i) y[i] = 0;
    for (int r = 0; r < n; ++r)
        for (int k = m.ia[r]; k < m.ia[r + 1]; ++k) {
            int c = m.ja[k];
            y[r] += a[k] * x[c];
            if (c != r) y[c] += a[k] * x[r];
        }
}
void orc_inversion_step(const orc_mesh* h, const double* p, double slackness, double* out) { inversionStep(h->m, p, slackness, out); }
double orc_filter_step_size(const orc_mesh* h, const double* p, double stepSize) { return filterStepSize(h->m, p, stepSize); }
}

// ---- element-sharded assembly (what each rank of a multi-GPU run computes; the ranks' results are summed by an
// all-reduce).  Elasticity of the tets [t0, t1) only; the nodal terms (mass / DBC identity on the diagonal, inertia
// part of the gradient, Optimizer.cpp:3438-3450, 3638-3668) are contributed by the rank with owner != 0.
extern "C" void orc_assemble_shard(const orc_mesh* h, double coef, int projectDBC, int t0, int t1, int owner,
    const double* xTilde_colmajor, double* a, double* grad)
{
    const Mesh& m = h->m;
    const size_t nnz = m.ja.size();
    for (size_t i = 0; i < nnz; ++i) a[i] = 0;
    for (int i = 0; i < 3 * m.nV; ++i) grad[i] = 0;
    for (int t = t0; t < t1; ++t) {
        double H[144], g[12];
        int vInd[4];
        elemHessian(m, t, coef, true, H);
        elemGradient(m, t, coef, g);
        for (int k = 0; k < 4; ++k) {
            int v = m.Fi(t, k);
            vInd[k] = m.isProjectDBC(v, projectDBC != 0) ? (-v - 1) : v;
            if (!(projectDBC && m.dbcType[v] != 0))
                for (int c = 0; c < 3; ++c) grad[3 * v + c] += g[3 * k + c];
        }
        for (int k = 0; k < 4; ++k)
            if (vInd[k] >= 0) addBlockToMatrix(m, a, H, vInd, k); // identity rows come from the owner below
    }
    if (owner) {
        for (int v = 0; v < m.nV; ++v) {
            const bool proj = m.isProjectDBC(v, projectDBC != 0);
            for (int c = 0; c < 3; ++c) {
                int k = m.findEntry(3 * v + c, 3 * v + c);
                if (proj) a[k] = 1.0;
                else {
                    a[k] += m.mass[v];
                    grad[3 * v + c] += m.mass[v] * (m.V[v + m.nV * c] - xTilde_colmajor[v + m.nV * c]);
                }
            }
        }
    }
}
