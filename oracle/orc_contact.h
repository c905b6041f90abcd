// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_api.h).
// Contact half of the hot path: closest-feature distances with first and second derivatives, the C2 clamped log
// barrier, constraint typing and constraint-set construction, barrier energy / gradient / Hessian assembly.
#pragma once
#include "orc_core.h"
#include <array>
#include <vector>

namespace orc {

typedef std::array<int, 4> MMCVID; // MeshCollisionUtils.hpp:24-118 (same sign encoding)

enum StencilKind { K_PP = 0, K_PE = 1, K_PT = 2, K_EE = 3 };

// squared distance d of the stencil, gradient g (3*n entries) and Hessian H (12 x 12 column-major, leading 3n x 3n used)
//   PP  |v0 - v1|^2                                  MeshCollisionUtils.hpp:156-176
//   PE  |(v1-v0) x (v2-v0)|^2 / |v2-v1|^2            :227-631
//   PT  ((v0-v1).n)^2 / |n|^2,  n = (v2-v1)x(v3-v1)  :685-1230
//   EE  ((v2-v0).n)^2 / |n|^2,  n = (v1-v0)x(v3-v2)  :1287-2015
// The reference differentiates these with MATLAB-generated straight-line code; here the same functions are
// differentiated in vector form (triple product s = w.(e x f), q = |e x f|^2, chain rule through the +-1
// maps from node coordinates to w, e, f), which agrees to round-off.
void stencil_distance(int kind, const double X[4][3], double* d, double* g, double* H);
int stencil_nodes(int kind);
// c = |(v1-v0) x (v3-v2)|^2 with derivatives (computeEECrossSqNorm*, MeshCollisionUtils.hpp:2409-2777)
void cross_sqnorm(const double X[4][3], double* c, double* g, double* H);
// BarrierFunctions.hpp:56-83 (BARRIER_FUNC_TYPE 2)
void barrier(double d, double dHat, double* b, double* gb, double* Hb);
// mollifier e(c), e'(c), e''(c)  (MeshCollisionUtils.hpp:2834-2866); zero derivatives beyond eps_x
void mollifier(double c, double eps_x, double* e, double* eg, double* eH);
int dType_PT(const double v0[3], const double v1[3], const double v2[3], const double v3[3]); // :2160-2210
int dType_EE(const double v0[3], const double v1[3], const double v2[3], const double v3[3]); // :2073-2158

struct ContactSets {
    std::vector<MMCVID> active; // MMActiveSet.back()
    std::vector<MMCVID> paraEE; // paraEEMMCVIDSet.back()
    std::vector<std::array<int, 2>> paraEEeIeJ; // paraEEeIeJSet.back()
    std::vector<std::array<int, 2>> csPTEE; // MMActiveSet_CCD.back(): (-svI-1, sfI) or (eI, eJ)
};

// SelfCollisionHandler::computeConstraintSet (SelfCollisionHandler.cpp:2149-2478); brute == true scans all pairs
// (the reference's non-USE_SH_CCS branch), otherwise a uniform grid prunes candidates (SpatialHash.hpp role).
void computeConstraintSet(const Mesh& m, double dHat, bool brute, ContactSets& out);
// kappa * sum mult * b(d) + kappa * sum e * b(d)   (Optimizer.cpp:3252-3353)
double contactEnergy(const Mesh& m, const ContactSets& cs, double dHat, double kappa);
// grad += kappa (...)   (Optimizer.cpp:3463-3517, SelfCollisionHandler.cpp:84-148, 2990-3036); DBC rows zeroed after
void contactGradient(const Mesh& m, const ContactSets& cs, double dHat, double kappa, bool projectDBC, double* grad);
// a += PSD-projected barrier Hessians (SelfCollisionHandler.cpp:418-561, 3039-3201)
void contactHessian(const Mesh& m, const ContactSets& cs, double dHat, double kappa, bool projectDBC, double* a);
// augmentConnectivity (SelfCollisionHandler.cpp:330-415): node pairs the barrier Hessians couple
void contactConnectivity(const Mesh& m, const ContactSets& cs, std::vector<std::pair<int, int>>& pairs);

// conservative CCD (contract: every tested pair keeps >= (1 - slackness) of its current distance); see orc_contact.cpp
double accd(int kind, const double X[4][3], const double P[4][3], double eta, double tmax);
double accdSmall(int n, const double X[3][3], const double P[3][3], double eta, double tmax);
double fullCcdReference(const Mesh& m, const double* p, double slackness, double alpha, double* alphaCapped, int arg[3], int* nCand);
double ccdStepBound(const Mesh& m, const std::vector<std::array<int, 2>>& pairs, const double* p, double slackness, double stepSize,
    int* argPair);
void sweptCandidates(const Mesh& m, const double* p, double stepSize, std::vector<std::array<int, 2>>& out);
bool isIntersected(const Mesh& m); // any surface edge through any surface triangle (SelfCollisionHandler.cpp:3255-3300)

// ---- analytic half-space obstacle (HalfSpace.cpp:41-85): points x with n.x + D > 0 are outside, constraint d = (n.x + D)^2
struct HalfSpace {
    double n[3], D, o[3];
    void init(const double origin[3], const double normal[3]);
    double dist(const Mesh& m, int v) const { return n[0] * m.Vx(v, 0) + n[1] * m.Vx(v, 1) + n[2] * m.Vx(v, 2) + D; }
};
void hsConstraintSet(const Mesh& m, const HalfSpace& h, double dHat, std::vector<int>& set); // CollisionObject.h:323-351
double hsEnergy(const Mesh& m, const HalfSpace& h, const std::vector<int>& set, double dHat, double kappa); // Optimizer.cpp:3254-3267
void hsGradient(const Mesh& m, const HalfSpace& h, const std::vector<int>& set, double dHat, double kappa, double* grad); // HalfSpace.cpp:121-143
void hsHessian(const Mesh& m, const HalfSpace& h, const std::vector<int>& set, double dHat, double kappa, bool projectDBC, double* a); // :169-214
double hsStepBound(const Mesh& m, const HalfSpace& h, const double* p, double slackness, double stepSize); // :242-269
double hsMove(const Mesh& m, HalfSpace& h, const double delta[3], double slackness); // HalfSpace::move, :389-416; returns the fraction left
bool hsIntersected(const Mesh& m, const HalfSpace& h); // CollisionObject.h:386-401 (fires only on d == 0: d is a square)

} // namespace orc

namespace orc {
// ---- lagged friction: FrictionUtils.hpp:24-347, SelfCollisionHandler.cpp:2481-2988, HalfSpace.cpp:272-381
struct FrictionLag {
    std::vector<MMCVID> set; // MMActiveSet_lastH
    std::vector<double> lambda; // MMLambda_lastH: -kappa b'(d) 2 sqrt(d), times the PP / PE multiplicity
    std::vector<std::array<double, 2>> coord; // MMDistCoord: closest-point parameters
    std::vector<std::array<double, 6>> basis; // MMTanBasis: 3 x 2, column-major
};
// Optimizer.cpp:1578-1598: multipliers + computeDistCoordAndTanBasis at the current positions
void frictionLagUpdate(const Mesh& m, const std::vector<MMCVID>& active, double dHat, double kappa, FrictionLag& lag);
double frictionEnergy(const Mesh& m, const double* Vt_colmajor, const FrictionLag& lag, double eps2, double coef);
void frictionGradient(const Mesh& m, const double* Vt_colmajor, const FrictionLag& lag, double eps2, double coef, double* grad);
void frictionHessian(const Mesh& m, const double* Vt_colmajor, const FrictionLag& lag, double eps2, double coef, bool projectDBC, double* a);
void frictionConnectivity(const FrictionLag& lag, std::vector<std::pair<int, int>>& pairs); // augmentConnectivity on the lagged set
// half-space (C0 clamping, HalfSpace.cpp:272-381); lambda per vertex of `set`
void hsFrictionLagUpdate(const Mesh& m, const HalfSpace& h, const std::vector<int>& set, double dHat, double kappa, std::vector<double>& lambda);
double hsFrictionEnergy(const Mesh& m, const double* Vt, const HalfSpace& h, const std::vector<int>& set, const std::vector<double>& lambda,
    double mu, double eps2);
void hsFrictionGradient(const Mesh& m, const double* Vt, const HalfSpace& h, const std::vector<int>& set, const std::vector<double>& lambda,
    double mu, double eps2, double* grad);
void hsFrictionHessian(const Mesh& m, const double* Vt, const HalfSpace& h, const std::vector<int>& set, const std::vector<double>& lambda,
    double mu, double eps2, bool projectDBC, double* a);
} // namespace orc
