// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_api.h).
//
// Lagged smoothed Coulomb friction, restated from
//   tangent bases / closest points / relative sliding / T^T T     src/CollisionObject/FrictionUtils.hpp:24-262
//   C1 static-friction clamping f0, f1/|u|, f2                     FrictionUtils.hpp:280-294 (SFCLAMPING_ORDER 1, Types.hpp:42)
//   multipliers + computeDistCoordAndTanBasis                      Optimizer.cpp:1578-1598, SelfCollisionHandler.cpp:2481-2527
//   computeFrictionEnergy / augmentFrictionGradient / ...Hessian   SelfCollisionHandler.cpp:2530-2988
//   half-space friction (C0 clamping)                              src/CollisionObject/HalfSpace.cpp:272-381
// All four stencil kinds share one form: with node weights wt_k (FrictionUtils' lift / TTT coefficients) and the 3 x 2
// tangent basis B, T^T = [wt_k B^T]_k, u = B^T sum_k wt_k (x_k - x_k^t).
#include "orc_api.h"
#include "orc_contact.h"
#include "orc_math.h"
#include <cmath>

namespace orc {

namespace {
struct FStencil {
    int kind, n, node[4];
    double mult;
};
FStencil decodeF(const MMCVID& c)
{
    FStencil s;
    s.mult = 1.0;
    if (c[0] >= 0) {
        s.kind = K_EE;
        s.n = 4;
        for (int i = 0; i < 4; ++i) s.node[i] = c[i];
        return s;
    }
    s.node[0] = -c[0] - 1;
    s.node[1] = c[1];
    if (c[2] < 0) {
        s.kind = K_PP;
        s.n = 2;
        s.mult = -c[3];
    }
    else if (c[3] < 0) {
        s.kind = K_PE;
        s.n = 3;
        s.node[2] = c[2];
        s.mult = -c[3];
    }
    else {
        s.kind = K_PT;
        s.n = 4;
        s.node[2] = c[2];
        s.node[3] = c[3];
    }
    return s;
}
void sub(const double* a, const double* b, double* c)
{
    for (int i = 0; i < 3; ++i) c[i] = a[i] - b[i];
}
double dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
void cross(const double* a, const double* b, double* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
void normalize(double* a)
{
    const double l = std::sqrt(dot(a, a));
    for (int i = 0; i < 3; ++i) a[i] /= l;
}
// 2x2 SPD solve (Eigen's ldlt().solve in FrictionUtils.hpp:45,118)
void solve2(double a, double b, double d, double r0, double r1, double* x)
{
    const double l10 = b / a, d1 = d - l10 * b;
    const double y1 = r1 - l10 * r0;
    x[1] = y1 / d1;
    x[0] = r0 / a - l10 * x[1];
}
void weights(int kind, const double* coord, double* wt)
{
    if (kind == K_PT) {
        wt[0] = 1.0;
        wt[1] = -1.0 + coord[0] + coord[1];
        wt[2] = -coord[0];
        wt[3] = -coord[1];
    }
    else if (kind == K_EE) {
        wt[0] = 1.0 - coord[0];
        wt[1] = coord[0];
        wt[2] = coord[1] - 1.0;
        wt[3] = -coord[1];
    }
    else if (kind == K_PE) {
        wt[0] = 1.0;
        wt[1] = coord[0] - 1.0;
        wt[2] = -coord[0];
        wt[3] = 0.0;
    }
    else {
        wt[0] = 1.0;
        wt[1] = -1.0;
        wt[2] = wt[3] = 0.0;
    }
}
// u = B^T sum_k wt_k (x_k - x_k^t)
void slide(const Mesh& m, const double* Vt, const FStencil& s, const double* wt, const double* B, double* u)
{
    double r[3] = { 0, 0, 0 };
    for (int k = 0; k < s.n; ++k)
        for (int c = 0; c < 3; ++c) r[c] += wt[k] * (m.Vx(s.node[k], c) - Vt[s.node[k] + m.nV * c]);
    u[0] = B[0] * r[0] + B[1] * r[1] + B[2] * r[2];
    u[1] = B[3] * r[0] + B[4] * r[1] + B[5] * r[2];
}
} // namespace

void frictionLagUpdate(const Mesh& m, const std::vector<MMCVID>& active, double dHat, double kappa, FrictionLag& lag)
{
    lag.set = active;
    const size_t n = active.size();
    lag.lambda.assign(n, 0.0);
    lag.coord.assign(n, { 0.0, 0.0 });
    lag.basis.assign(n, { 0, 0, 0, 0, 0, 0 });
    for (size_t i = 0; i < n; ++i) {
        const FStencil s = decodeF(active[i]);
        double X[4][3] = { { 0 } };
        for (int k = 0; k < s.n; ++k)
            for (int c = 0; c < 3; ++c) X[k][c] = m.Vx(s.node[k], c);
        double d, b, gb, Hb;
        stencil_distance(s.kind, X, &d, nullptr, nullptr);
        barrier(d, dHat, &b, &gb, &Hb);
        lag.lambda[i] = gb * (-kappa * 2.0 * std::sqrt(d)); // Optimizer.cpp:1586-1587
        if (active[i][3] < -1) lag.lambda[i] *= -active[i][3]; // PP / PE duplication (:1588-1591)
        double* B = lag.basis[i].data();
        double* co = lag.coord[i].data();
        double t0[3], t1[3], tmp[3];
        if (s.kind == K_EE) {
            double e20[3], e01[3], e23[3];
            sub(X[0], X[2], e20);
            sub(X[1], X[0], e01);
            sub(X[3], X[2], e23);
            solve2(dot(e01, e01), -dot(e23, e01), dot(e23, e23), -dot(e20, e01), dot(e20, e23), co); // computeClosestPoint_EE
            for (int c = 0; c < 3; ++c) t0[c] = e01[c];
            cross(e01, e23, tmp);
            cross(tmp, e01, t1);
        }
        else if (s.kind == K_PT) {
            double e1[3], e2[3], w[3];
            sub(X[2], X[1], e1);
            sub(X[3], X[1], e2);
            sub(X[0], X[1], w);
            solve2(dot(e1, e1), dot(e1, e2), dot(e2, e2), dot(e1, w), dot(e2, w), co); // computeClosestPoint_PT
            for (int c = 0; c < 3; ++c) t0[c] = e1[c];
            cross(e1, e2, tmp);
            cross(tmp, e1, t1);
        }
        else if (s.kind == K_PE) {
            double e12[3], w[3];
            sub(X[2], X[1], e12);
            sub(X[0], X[1], w);
            co[0] = dot(w, e12) / dot(e12, e12);
            for (int c = 0; c < 3; ++c) t0[c] = e12[c];
            cross(e12, w, t1);
        }
        else { // PP: FrictionUtils.hpp:229-243
            double v01[3], xC[3], yC[3];
            sub(X[1], X[0], v01);
            const double ex[3] = { 1, 0, 0 }, ey[3] = { 0, 1, 0 };
            cross(ex, v01, xC);
            cross(ey, v01, yC);
            const double* pick = dot(xC, xC) > dot(yC, yC) ? xC : yC;
            for (int c = 0; c < 3; ++c) t0[c] = pick[c];
            cross(v01, pick, t1);
        }
        normalize(t0);
        normalize(t1);
        for (int c = 0; c < 3; ++c) {
            B[c] = t0[c];
            B[3 + c] = t1[c];
        }
    }
}

double frictionEnergy(const Mesh& m, const double* Vt, const FrictionLag& lag, double eps2, double coef)
{
    const double eps = std::sqrt(eps2);
    double sum = 0;
    for (size_t i = 0; i < lag.set.size(); ++i) {
        const FStencil s = decodeF(lag.set[i]);
        double wt[4], u[2];
        weights(s.kind, lag.coord[i].data(), wt);
        slide(m, Vt, s, wt, lag.basis[i].data(), u);
        const double x2 = u[0] * u[0] + u[1] * u[1];
        if (x2 > eps2) sum += lag.lambda[i] * std::sqrt(x2);
        else sum += lag.lambda[i] * (x2 * (-std::sqrt(x2) / 3.0 + eps) / (eps * eps) + eps / 3.0); // f0_SF_C1
    }
    return sum * coef;
}

void frictionGradient(const Mesh& m, const double* Vt, const FrictionLag& lag, double eps2, double coef, double* grad)
{
    const double eps = std::sqrt(eps2);
    for (size_t i = 0; i < lag.set.size(); ++i) {
        const FStencil s = decodeF(lag.set[i]);
        const double* B = lag.basis[i].data();
        double wt[4], u[2];
        weights(s.kind, lag.coord[i].data(), wt);
        slide(m, Vt, s, wt, B, u);
        const double x2 = u[0] * u[0] + u[1] * u[1];
        const double sc = (x2 > eps2) ? 1.0 / std::sqrt(x2) : (-std::sqrt(x2) + 2.0 * eps) / (eps * eps); // f1 / |u|
        double t3[3];
        for (int c = 0; c < 3; ++c) t3[c] = B[c] * (u[0] * sc) + B[3 + c] * (u[1] * sc);
        for (int k = 0; k < s.n; ++k)
            for (int c = 0; c < 3; ++c) grad[3 * s.node[k] + c] += coef * lag.lambda[i] * wt[k] * t3[c];
    }
}

void frictionHessian(const Mesh& m, const double* Vt, const FrictionLag& lag, double eps2, double coef, bool projectDBC, double* a)
{
    const double eps = std::sqrt(eps2);
    for (size_t i = 0; i < lag.set.size(); ++i) {
        const FStencil s = decodeF(lag.set[i]);
        const double* B = lag.basis[i].data();
        double wt[4], u[2];
        weights(s.kind, lag.coord[i].data(), wt);
        slide(m, Vt, s, wt, B, u);
        const double x2 = u[0] * u[0] + u[1] * u[1], xn = std::sqrt(x2);
        const int n3 = 3 * s.n;
        double H[144], T[12]; // T = lift(u) = [wt_k B u]
        for (int k = 0; k < s.n; ++k)
            for (int c = 0; c < 3; ++c) T[3 * k + c] = wt[k] * (B[c] * u[0] + B[3 + c] * u[1]);
        double BBt[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) BBt[r + 3 * c] = B[r] * B[c] + B[3 + r] * B[3 + c];
        const double cl = coef * lag.lambda[i];
        double aI, bU;
        bool project;
        if (x2 > eps2) {
            aI = cl / xn;
            bU = -cl / (x2 * xn);
            project = true;
        }
        else {
            const double f1d = (-xn + 2.0 * eps) / (eps * eps), f2 = 2.0 * (eps - xn) / (eps * eps);
            aI = cl * f1d;
            project = (f2 != f1d) && x2 != 0.0;
            bU = project ? cl * (f2 - f1d) / x2 : 0.0;
        }
        for (int k = 0; k < s.n; ++k)
            for (int l = 0; l < s.n; ++l)
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c)
                        H[(3 * k + r) + n3 * (3 * l + c)] = aI * wt[k] * wt[l] * BBt[r + 3 * c] + bU * T[3 * k + r] * T[3 * l + c];
        if (project) make_pd(n3, H);
        for (int k = 0; k < s.n; ++k) {
            if (m.isProjectDBC(s.node[k], projectDBC)) continue;
            for (int l = 0; l < s.n; ++l) {
                if (m.isProjectDBC(s.node[l], projectDBC)) continue;
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) {
                        const int row = 3 * s.node[k] + r, col = 3 * s.node[l] + c;
                        if (row > col) continue; // addCoeff drops the lower triangle
                        a[m.findEntry(row, col)] += H[(3 * k + r) + n3 * (3 * l + c)];
                    }
            }
        }
    }
}

void frictionConnectivity(const FrictionLag& lag, std::vector<std::pair<int, int>>& pairs)
{
    auto link = [&](int x, int y) {
        if (x != y) pairs.push_back({ std::min(x, y), std::max(x, y) });
    };
    for (const auto& c : lag.set) { // SelfCollisionHandler.cpp:330-376
        const FStencil s = decodeF(c);
        if (s.kind == K_EE) {
            link(s.node[0], s.node[2]);
            link(s.node[0], s.node[3]);
            link(s.node[1], s.node[2]);
            link(s.node[1], s.node[3]);
        }
        else
            for (int k = 1; k < s.n; ++k) link(s.node[0], s.node[k]);
    }
}

// ---- half-space --------------------------------------------------------------------------------------------------
void hsFrictionLagUpdate(const Mesh& m, const HalfSpace& h, const std::vector<int>& set, double dHat, double kappa, std::vector<double>& lambda)
{
    lambda.assign(set.size(), 0.0);
    for (size_t i = 0; i < set.size(); ++i) {
        const double dist = h.dist(m, set[i]), d = dist * dist;
        double b, gb, Hb;
        barrier(d, dHat, &b, &gb, &Hb);
        lambda[i] = gb * (-kappa * 2.0 * std::sqrt(d)); // Optimizer.cpp:1563-1566
    }
}
static void hsProj(const Mesh& m, const double* Vt, const HalfSpace& h, int v, double* vp)
{
    double vd[3];
    for (int c = 0; c < 3; ++c) vd[c] = m.Vx(v, c) - Vt[v + m.nV * c];
    const double dn = vd[0] * h.n[0] + vd[1] * h.n[1] + vd[2] * h.n[2];
    for (int c = 0; c < 3; ++c) vp[c] = vd[c] - dn * h.n[c];
}
double hsFrictionEnergy(const Mesh& m, const double* Vt, const HalfSpace& h, const std::vector<int>& set, const std::vector<double>& lambda,
    double mu, double eps2)
{
    const double eps = std::sqrt(eps2);
    double Ef = 0;
    for (size_t i = 0; i < set.size(); ++i) {
        double vp[3];
        hsProj(m, Vt, h, set[i], vp);
        const double m2 = dot(vp, vp);
        if (m2 > eps2) Ef += mu * lambda[i] * (std::sqrt(m2) - eps * 0.5);
        else Ef += mu * lambda[i] * m2 / eps * 0.5;
    }
    return Ef;
}
void hsFrictionGradient(const Mesh& m, const double* Vt, const HalfSpace& h, const std::vector<int>& set, const std::vector<double>& lambda,
    double mu, double eps2, double* grad)
{
    const double eps = std::sqrt(eps2);
    for (size_t i = 0; i < set.size(); ++i) {
        double vp[3];
        hsProj(m, Vt, h, set[i], vp);
        const double m2 = dot(vp, vp);
        const double sc = (m2 > eps2) ? mu * lambda[i] / std::sqrt(m2) : mu * lambda[i] / eps;
        for (int c = 0; c < 3; ++c) grad[3 * set[i] + c] += sc * vp[c];
    }
}
void hsFrictionHessian(const Mesh& m, const double* Vt, const HalfSpace& h, const std::vector<int>& set, const std::vector<double>& lambda,
    double mu, double eps2, bool projectDBC, double* a)
{
    const double eps = std::sqrt(eps2);
    for (size_t i = 0; i < set.size(); ++i) {
        const int v = set[i];
        if (projectDBC && m.isDBC(v)) continue;
        const double ml = mu * lambda[i];
        double vp[3], H[9];
        hsProj(m, Vt, h, v, vp);
        const double m2 = dot(vp, vp);
        if (m2 > eps2) {
            const double mag = std::sqrt(m2);
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c)
                    H[r + 3 * c] = vp[r] * (-ml / m2 / mag) * vp[c] + ((r == c ? 1.0 : 0.0) - h.n[r] * h.n[c]) * (ml / mag);
            make_pd(3, H);
        }
        else
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) H[r + 3 * c] = ((r == c ? 1.0 : 0.0) - h.n[r] * h.n[c]) * (ml / eps);
        for (int r = 0; r < 3; ++r)
            for (int c = r; c < 3; ++c) a[m.findEntry(3 * v + r, 3 * v + c)] += H[r + 3 * c];
    }
}

} // namespace orc

using namespace orc;

extern "C" {
struct orc_friction {
    FrictionLag lag;
};
orc_friction* orc_friction_create() { return new orc_friction; }
void orc_friction_destroy(orc_friction* f) { delete f; }
// lag the given active set (n x 4 MMCVID tuples) at the mesh's current positions
void orc_friction_update(orc_friction* f, const orc_mesh* mh, int n, const int* active4, double dHat, double kappa)
{
    std::vector<MMCVID> act(n);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 4; ++k) act[i][k] = active4[4 * i + k];
    frictionLagUpdate(mh->m, act, dHat, kappa, f->lag);
}
void orc_friction_get(const orc_friction* f, double* lambda, double* coord2, double* basis6)
{
    for (size_t i = 0; i < f->lag.set.size(); ++i) {
        lambda[i] = f->lag.lambda[i];
        for (int k = 0; k < 2; ++k) coord2[2 * i + k] = f->lag.coord[i][k];
        for (int k = 0; k < 6; ++k) basis6[6 * i + k] = f->lag.basis[i][k];
    }
}
double orc_friction_energy(const orc_friction* f, const orc_mesh* mh, const double* Vt, double eps2, double coef)
{
    return frictionEnergy(mh->m, Vt, f->lag, eps2, coef);
}
void orc_friction_gradient(const orc_friction* f, const orc_mesh* mh, const double* Vt, double eps2, double coef, double* grad)
{
    frictionGradient(mh->m, Vt, f->lag, eps2, coef, grad);
}
void orc_friction_hessian(const orc_friction* f, const orc_mesh* mh, const double* Vt, double eps2, double coef, int projectDBC, double* a)
{
    frictionHessian(mh->m, Vt, f->lag, eps2, coef, projectDBC != 0, a);
}
}
