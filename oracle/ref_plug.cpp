// ORACLE / TEST INFRASTRUCTURE -- the three things oracle/_ref/libipcref.so cannot take from /root/reference because
// they live in un-vendored dependencies, plugged with what this repository already states elsewhere:
//
//  1. CTCD::vertexFaceCTCD / edgeEdgeCTCD / vertexEdgeCTCD / vertexVertexCTCD  (CCD-Wrapper@23907da)
//       -> the per-pair conservative advancement of oracle/orc_contact.cpp (`orc_accd`), i.e. "CCD by contract"
//          (DESIGN.md section 2).  eta arrives as an absolute distance ((1 - slackness) * current distance at the
//          reference's call sites); the advancement takes it as a fraction of the current distance.  eta = 0 (the
//          reference's retry when t < 1e-6) maps to the fraction 0.01, as in oracle/orc_contact.cpp::ccdStepBound.
//  2. LinSysSolver<...>::create  (CHOLMOD through SuiteSparse, a system package)
//       -> a subclass of the reference's own LinSysSolver.hpp base (its set_pattern / addCoeff / setCoeff / IJ2aI are
//          the real code) whose analyze / factorize / solve call the oracle's multifrontal Cholesky (orc_chol.cpp).
//          factorize() returns false exactly when a non-positive pivot is met, as CHOLMODSolver.cpp:123-154 does.
//  3. main()  -> exported as ipcref_main(argc, argv) so that a test can run a scene script through the reference's own
//          main.cpp / Config.cpp / Optimizer.cpp in offline mode (progMode 100).
//
// Everything else in libipcref.so is compiled from the reference's sources where they lie (see Makefile.ref).
#include "CTCD.h"
#include "LinSysSolver.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" {
// oracle/orc_api.h (liborc.so)
double orc_accd(int kind, const double* X12, const double* P12, double eta, double tmax);
typedef struct orc_chol orc_chol;
orc_chol* orc_chol_create(int n, const int* ia, const int* ja, int nthreads);
void orc_chol_destroy(orc_chol*);
int orc_chol_factorize(orc_chol*, const double* a);
void orc_chol_solve(const orc_chol*, const double* rhs, double* x);
}

namespace {
enum { K_PP = 0,
    K_PE = 1,
    K_PT = 2,
    K_EE = 3 }; // oracle/orc_contact.h

void pack(const Eigen::Vector3d* s, const Eigen::Vector3d* e, int n, double* X, double* P)
{
    for (int k = 0; k < 4; ++k)
        for (int c = 0; c < 3; ++c) {
            const int kk = k < n ? k : n - 1;
            X[3 * k + c] = s[kk][c];
            P[3 * k + c] = e[kk][c] - s[kk][c];
        }
}
// additive advancement for a stencil kind the oracle's accd() covers through PT (a degenerate triangle has the
// distance of its edge / point): PP as PT(p; q,q,q), PE as PT(p; a,b,b)
bool advance(int kind, const double* X, const double* P, double d0, double eta, double& t)
{
    const double frac = (eta > 0.0 && d0 > 0.0) ? eta / d0 : 0.01;
    const double toc = orc_accd(kind, X, P, frac, 1.0);
    if (std::getenv("IPCREF_LOG_CCD")) std::fprintf(stderr, "ccd kind %d d0 %g eta %g frac %g -> %g\n", kind, d0, eta, frac, toc);
    if (toc < 1.0) {
        t = toc;
        return true;
    }
    return false;
}
} // namespace

extern "C" double orc_unclassified_distance(int kind, const double* X12); // liborc.so

bool CTCD::vertexFaceCTCD(const Eigen::Vector3d& q0s, const Eigen::Vector3d& q1s, const Eigen::Vector3d& q2s, const Eigen::Vector3d& q3s,
    const Eigen::Vector3d& q0e, const Eigen::Vector3d& q1e, const Eigen::Vector3d& q2e, const Eigen::Vector3d& q3e, double eta, double& t)
{
    const Eigen::Vector3d s[4] = { q0s, q1s, q2s, q3s }, e[4] = { q0e, q1e, q2e, q3e };
    double X[12], P[12];
    pack(s, e, 4, X, P);
    return advance(K_PT, X, P, orc_unclassified_distance(K_PT, X), eta, t);
}
bool CTCD::edgeEdgeCTCD(const Eigen::Vector3d& q0s, const Eigen::Vector3d& p0s, const Eigen::Vector3d& q1s, const Eigen::Vector3d& p1s,
    const Eigen::Vector3d& q0e, const Eigen::Vector3d& p0e, const Eigen::Vector3d& q1e, const Eigen::Vector3d& p1e, double eta, double& t)
{
    const Eigen::Vector3d s[4] = { q0s, p0s, q1s, p1s }, e[4] = { q0e, p0e, q1e, p1e };
    double X[12], P[12];
    pack(s, e, 4, X, P);
    return advance(K_EE, X, P, orc_unclassified_distance(K_EE, X), eta, t);
}
extern "C" double orc_accd_small(int n, const double* X9, const double* P9, double eta, double tmax); // liborc.so
namespace {
// point-point and point-segment pairs (the reference's full CCD also sweeps those, SelfCollisionHandler.cpp:1011-1100): the same
// additive advancement on the point-point / point-segment distance (oracle/orc_contact.cpp::accdSmall)
bool advanceSmall(int n, double X[3][3], const double P[3][3], double eta, double& t)
{
    double d2 = 0;
    if (n == 2) {
        for (int c = 0; c < 3; ++c) d2 += (X[0][c] - X[1][c]) * (X[0][c] - X[1][c]);
    }
    const double frac = (eta > 0.0) ? -1.0 : 0.01;
    (void)d2;
    double X9[9], P9[9];
    for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 3; ++c) {
            const int kk = k < n ? k : n - 1;
            X9[3 * k + c] = X[kk][c];
            P9[3 * k + c] = P[kk][c];
        }
    // eta arrives as an absolute distance: the fraction of the current distance is recovered by one call with eta = 0 ... the
    // current distance is what accdSmall starts from, so pass eta / d0 computed here
    double d0;
    {
        double ab[3], ap[3], abab = 0, apab = 0;
        const double* pnt = X9;
        const double* a = X9 + 3;
        const double* b = X9 + (n == 2 ? 3 : 6);
        for (int c = 0; c < 3; ++c) {
            ab[c] = b[c] - a[c];
            ap[c] = pnt[c] - a[c];
            abab += ab[c] * ab[c];
            apab += ap[c] * ab[c];
        }
        double s_ = abab > 0.0 ? apab / abab : 0.0;
        s_ = s_ < 0.0 ? 0.0 : (s_ > 1.0 ? 1.0 : s_);
        double q = 0;
        for (int c = 0; c < 3; ++c) {
            const double r = ap[c] - s_ * ab[c];
            q += r * r;
        }
        d0 = std::sqrt(q);
    }
    const double f = (frac > 0.0 || d0 <= 0.0) ? 0.01 : eta / d0;
    const double toc = orc_accd_small(n, X9, P9, f, 1.0);
    if (std::getenv("IPCREF_LOG_CCD")) std::fprintf(stderr, "ccd small n %d eta %g frac %g -> %g\n", n, eta, f, toc);
    if (toc < 1.0) {
        t = toc;
        return true;
    }
    return false;
}
} // namespace
bool CTCD::vertexEdgeCTCD(const Eigen::Vector3d& q0s, const Eigen::Vector3d& q1s, const Eigen::Vector3d& q2s,
    const Eigen::Vector3d& q0e, const Eigen::Vector3d& q1e, const Eigen::Vector3d& q2e, double eta, double& t)
{
    double X[3][3], P[3][3];
    const Eigen::Vector3d s[3] = { q0s, q1s, q2s }, e[3] = { q0e, q1e, q2e };
    for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 3; ++c) {
            X[k][c] = s[k][c];
            P[k][c] = e[k][c] - s[k][c];
        }
    return advanceSmall(3, X, P, eta, t);
}
bool CTCD::vertexVertexCTCD(const Eigen::Vector3d& q1s, const Eigen::Vector3d& q2s,
    const Eigen::Vector3d& q1e, const Eigen::Vector3d& q2e, double eta, double& t)
{
    double X[3][3], P[3][3];
    const Eigen::Vector3d s[2] = { q1s, q2s }, e[2] = { q1e, q2e };
    for (int k = 0; k < 2; ++k)
        for (int c = 0; c < 3; ++c) {
            X[k][c] = s[k][c];
            P[k][c] = e[k][c] - s[k][c];
        }
    return advanceSmall(2, X, P, eta, t);
}

#ifndef IPCREF_PLUG_CTCD_ONLY // tests/adapters/: the executable that runs the reference's main() on the HIP adapters brings its own factory and main
namespace IPC {

template <typename vectorTypeI, typename vectorTypeS>
class RefPlugSolver : public LinSysSolver<vectorTypeI, vectorTypeS> {
    typedef LinSysSolver<vectorTypeI, vectorTypeS> Base;
    orc_chol* h_ = nullptr;
    std::vector<int> ia0_, ja0_;

public:
    ~RefPlugSolver() override
    {
        if (h_) orc_chol_destroy(h_);
    }
    LinSysSolverType type() const override { return LinSysSolverType::CHOLMOD; }
    void analyze_pattern(void) override
    {
        // the base class keeps ia / ja 1-based (LinSysSolver.hpp:146)
        const int n = Base::numRows;
        ia0_.resize((size_t)n + 1);
        for (int i = 0; i <= n; ++i) ia0_[(size_t)i] = Base::ia[i] - 1;
        ja0_.resize((size_t)Base::ja.size());
        for (size_t k = 0; k < ja0_.size(); ++k) ja0_[k] = Base::ja[(Eigen::Index)k] - 1;
        if (h_) orc_chol_destroy(h_);
        // IPCREF_THREADS (see refshim/tbb/parallel_for.h): the timed `cpu_reference` of bench.py runs the factorisation on the same number of threads
        const char* e = std::getenv("IPCREF_THREADS");
        h_ = orc_chol_create(n, ia0_.data(), ja0_.data(), e && std::atoi(e) > 1 ? std::atoi(e) : 1);
    }
    bool factorize(void) override { return orc_chol_factorize(h_, Base::a.data()) == 1; }
    void solve(Eigen::VectorXd& rhs, Eigen::VectorXd& result) override
    {
        result.resize(rhs.size());
        orc_chol_solve(h_, rhs.data(), result.data());
    }
};

template <typename vectorTypeI, typename vectorTypeS>
LinSysSolver<vectorTypeI, vectorTypeS>* LinSysSolver<vectorTypeI, vectorTypeS>::create(const LinSysSolverType)
{
    return new RefPlugSolver<vectorTypeI, vectorTypeS>();
}
template class LinSysSolver<Eigen::VectorXi, Eigen::VectorXd>;

} // namespace IPC

int ipc_reference_main(int argc, char* argv[]); // main.cpp compiled with -Dmain=ipc_reference_main

extern "C" int ipcref_main(int argc, char** argv) { return ipc_reference_main(argc, argv); }
#endif // IPCREF_PLUG_CTCD_ONLY
