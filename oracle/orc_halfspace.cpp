// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_api.h).
//
// Analytic half-space obstacle, restated from
//   HalfSpace::init                              src/CollisionObject/HalfSpace.cpp:41-85   (n normalised, D = -n.origin)
//   CollisionObject::computeConstraintSet        src/CollisionObject/CollisionObject.h:323-351
//   HalfSpace::evaluateConstraint                HalfSpace.cpp:106-111                      d = (n.x + D)^2
//   HalfSpace::leftMultiplyConstraintJacobianT   HalfSpace.cpp:121-143                      grad += kappa b'(d) 2 dist n
//   HalfSpace::augmentIPHessian                  HalfSpace.cpp:169-214                      kappa (4 b'' d + 2 b') n n^T if positive
//   HalfSpace::largestFeasibleStepSize           HalfSpace.cpp:242-269                      ray bound * slackness
//   CollisionObject::isIntersected               CollisionObject.h:386-401
#include "orc_api.h"
#include "orc_contact.h"
#include <cmath>

namespace orc {

void HalfSpace::init(const double origin[3], const double normal[3])
{
    const double len = std::sqrt(normal[0] * normal[0] + normal[1] * normal[1] + normal[2] * normal[2]);
    for (int c = 0; c < 3; ++c) n[c] = normal[c] / len;
    for (int c = 0; c < 3; ++c) o[c] = origin[c];
    D = -(n[0] * origin[0] + n[1] * origin[1] + n[2] * origin[2]);
}

void hsConstraintSet(const Mesh& m, const HalfSpace& h, double dHat, std::vector<int>& set)
{
    set.clear();
    for (int v : m.SVI) { // ascending svI; every vertex of a volumetric mesh has codimension 3
        if (m.isDBC(v)) continue;
        const double dist = h.dist(m, v);
        if (dist * dist < dHat) set.push_back(v);
    }
}

double hsEnergy(const Mesh& m, const HalfSpace& h, const std::vector<int>& set, double dHat, double kappa)
{
    double sum = 0;
    for (int v : set) {
        const double dist = h.dist(m, v);
        double b, gb, Hb;
        barrier(dist * dist, dHat, &b, &gb, &Hb);
        sum += b;
    }
    return kappa * sum;
}

void hsGradient(const Mesh& m, const HalfSpace& h, const std::vector<int>& set, double dHat, double kappa, double* grad)
{
    for (int v : set) {
        const double dist = h.dist(m, v);
        double b, gb, Hb;
        barrier(dist * dist, dHat, &b, &gb, &Hb);
        for (int c = 0; c < 3; ++c) grad[3 * v + c] += kappa * gb * 2.0 * dist * h.n[c];
    }
}

void hsHessian(const Mesh& m, const HalfSpace& h, const std::vector<int>& set, double dHat, double kappa, bool projectDBC, double* a)
{
    for (int v : set) {
        if (m.isDBC(v) && projectDBC) continue;
        const double dist = h.dist(m, v), d = dist * dist;
        double b, gb, Hb;
        barrier(d, dHat, &b, &gb, &Hb);
        const double param = 4.0 * Hb * d + 2.0 * gb;
        if (!(param > 0.0)) continue;
        for (int r = 0; r < 3; ++r)
            for (int c = r; c < 3; ++c) // addCoeff drops the lower triangle (LinSysSolver.hpp:402-410)
                a[m.findEntry(3 * v + r, 3 * v + c)] += kappa * param * h.n[r] * h.n[c];
    }
}

double hsStepBound(const Mesh& m, const HalfSpace& h, const double* p, double slackness, double stepSize)
{
    double best = 1.0; // maxStepSizes[svI] starts at 1 (HalfSpace.cpp:256)
    for (int v : m.SVI) {
        if (m.isDBC(v)) continue;
        const double coef = h.n[0] * p[3 * v] + h.n[1] * p[3 * v + 1] + h.n[2] * p[3 * v + 2];
        if (coef < 0.0) best = std::min(best, -h.dist(m, v) / coef * slackness);
    }
    return std::min(stepSize, best);
}

double hsMove(const Mesh& m, HalfSpace& h, const double delta[3], double slackness)
{
    // HalfSpace.cpp:389-416: every surface node, Dirichlet or not, bounds the fraction of delta the plane may take
    const double coef = -(h.n[0] * delta[0] + h.n[1] * delta[1] + h.n[2] * delta[2]);
    double best = 1.0;
    if (coef < 0.0)
        for (int v : m.SVI) best = std::min(best, -h.dist(m, v) / coef * slackness);
    const double stepSize = std::min(1.0, best);
    const double origin[3] = { h.o[0] + stepSize * delta[0], h.o[1] + stepSize * delta[1], h.o[2] + stepSize * delta[2] };
    h.init(origin, h.n);
    return 1.0 - stepSize;
}

bool hsIntersected(const Mesh& m, const HalfSpace& h)
{
    for (int v = 0; v < m.nV; ++v)
        if (!m.isDBC(v)) {
            const double dist = h.dist(m, v);
            if (dist * dist <= 0.0) return true;
        }
    return false;
}

} // namespace orc

using namespace orc;

extern "C" {
struct orc_halfspace {
    HalfSpace h;
    std::vector<int> set;
};
orc_halfspace* orc_halfspace_create(const double* origin3, const double* normal3)
{
    orc_halfspace* o = new orc_halfspace;
    o->h.init(origin3, normal3);
    return o;
}
void orc_halfspace_destroy(orc_halfspace* o) { delete o; }
int orc_halfspace_build(orc_halfspace* o, const orc_mesh* mh, double dHat)
{
    hsConstraintSet(mh->m, o->h, dHat, o->set);
    return (int)o->set.size();
}
void orc_halfspace_get(const orc_halfspace* o, int* verts)
{
    for (size_t i = 0; i < o->set.size(); ++i) verts[i] = o->set[i];
}
double orc_halfspace_energy(const orc_halfspace* o, const orc_mesh* mh, double dHat, double kappa) { return hsEnergy(mh->m, o->h, o->set, dHat, kappa); }
void orc_halfspace_gradient(const orc_halfspace* o, const orc_mesh* mh, double dHat, double kappa, double* grad)
{
    hsGradient(mh->m, o->h, o->set, dHat, kappa, grad);
}
void orc_halfspace_hessian(const orc_halfspace* o, const orc_mesh* mh, double dHat, double kappa, int projectDBC, double* a)
{
    hsHessian(mh->m, o->h, o->set, dHat, kappa, projectDBC != 0, a);
}
double orc_halfspace_move(orc_halfspace* o, const orc_mesh* mh, const double* delta3, double slackness, double* originOut)
{
    const double left = hsMove(mh->m, o->h, delta3, slackness);
    for (int c = 0; c < 3; ++c) originOut[c] = o->h.o[c];
    return left;
}
double orc_halfspace_step_bound(const orc_halfspace* o, const orc_mesh* mh, const double* p, double slackness, double stepSize)
{
    return hsStepBound(mh->m, o->h, p, slackness, stepSize);
}
}
