// ORACLE / TEST INFRASTRUCTURE -- C entry points into the REFERENCE's own functions (compiled from /root/reference into
// oracle/_ref/libipcref.so, see Makefile.ref), so that tests/ and tools/make_golden_ref.py can evaluate single pieces
// of the Newton path on the same inputs as the oracle and the HIP library: Mesh<3> features, the elasticity energies
// (value / gradient / PSD-projected Hessian in the solver's CSR), the inversion filter, the spatial-hash constraint sets,
// barrier terms, step bounds, the intersection check.  No algorithm lives here: every function below only converts
// plain arrays to the reference's containers and calls the reference.
#include "Mesh.hpp"
#include "NeoHookeanEnergy.hpp"
#include "FixedCoRotEnergy.hpp"
#include "SelfCollisionHandler.hpp"
#include "HalfSpace.hpp"
#include "FrictionUtils.hpp"
#include "BarrierFunctions.hpp"
#include "IglUtils.hpp"
#include "get_feasible_steps.hpp"

#include <memory>
#include <vector>

using namespace IPC;

struct ipcref_mesh {
    std::unique_ptr<Mesh<3>> m;
    std::unique_ptr<Energy<3>> energy[2]; // 0 NH, 1 FCR
    std::vector<AutoFlipSVD<Eigen::Matrix3d>> svd;
    std::vector<Eigen::Matrix3d> F;
    // last constraint sets
    std::vector<MMCVID> active, para;
    std::vector<std::pair<int, int>> paraEIEJ, csPTEE;
    std::unique_ptr<LinSysSolver<Eigen::VectorXi, Eigen::VectorXd>> sol;
};

static Eigen::MatrixXd toMat(const double* p, int n, int c)
{
    Eigen::MatrixXd M(n, c);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < c; ++j) M(i, j) = p[(size_t)i * c + j];
    return M;
}
static Eigen::MatrixXi toMatI(const int* p, int n, int c)
{
    Eigen::MatrixXi M(n, c);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < c; ++j) M(i, j) = p[(size_t)i * c + j];
    return M;
}

extern "C" {

// V, T, SF row-major; components as main.cpp:880-1198 hands them over (node / surface-triangle ranges per loaded shape)
ipcref_mesh* ipcref_mesh_create(int nV, const double* V, int nT, const int* T, int nSF, const int* SF, int nComp, const int* nodeRange,
    const int* sfRange, double YM, double PR, double rho)
{
    ipcref_mesh* h = new ipcref_mesh;
    Eigen::MatrixXd Vm = toMat(V, nV, 3);
    Eigen::MatrixXi Tm = toMatI(T, nT, 4), SFm = toMatI(SF, nSF, 3), CE(0, 2);
    std::vector<int> nr(nodeRange, nodeRange + nComp + 1), sr(sfRange, sfRange + nComp + 1), cr((size_t)nComp + 1, 0), cd((size_t)nComp, 3);
    std::vector<std::pair<Eigen::Vector3i, Eigen::Vector3d>> none;
    std::vector<std::pair<Eigen::Vector3i, std::array<Eigen::Vector3d, 2>>> noneInit;
    h->m.reset(new Mesh<3>(Vm, Tm, SFm, CE, Vm, nr, sr, cr, cd, none, none, none, noneInit, {}, {}, {}, YM, PR, rho));
    h->m->resetDBCVertices(); // the constructor pins vertex 0 (2-D legacy, Mesh.cpp:417-423); AnimScripter::initAnimScript resets it too
    h->energy[0].reset(new NeoHookeanEnergy<3>());
    h->energy[1].reset(new FixedCoRotEnergy<3>());
    h->svd.resize((size_t)nT);
    h->F.resize((size_t)nT);
    return h;
}
void ipcref_mesh_destroy(ipcref_mesh* h) { delete h; }
void ipcref_mesh_set_positions(ipcref_mesh* h, const double* V)
{
    for (int i = 0; i < h->m->V.rows(); ++i)
        for (int j = 0; j < 3; ++j) h->m->V(i, j) = V[3 * i + j];
}
// type: 1 ZERO, 2 NONZERO (Mesh.hpp:41-45); replaces the current Dirichlet set
void ipcref_mesh_set_dbc(ipcref_mesh* h, int n, const int* ids, const int* types)
{
    std::map<int, DirichletBCType> mp;
    for (int i = 0; i < n; ++i) mp[ids[i]] = (DirichletBCType)types[i];
    h->m->resetDBCVertices(mp);
}
void ipcref_mesh_set_lame(ipcref_mesh* h, double YM, double PR) { h->m->setLameParam(YM, PR); }
void ipcref_mesh_features(const ipcref_mesh* h, double* restTriInv9, double* triArea, double* massDiag, double* mu, double* lam)
{
    const Mesh<3>& m = *h->m;
    for (int t = 0; t < m.F.rows(); ++t) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) restTriInv9[9 * t + 3 * i + j] = m.restTriInv[(size_t)t](i, j);
        triArea[t] = m.triArea[t];
        mu[t] = m.u[t];
        lam[t] = m.lambda[t];
    }
    Eigen::VectorXd d = m.massMatrix.diagonal();
    for (int v = 0; v < m.V.rows(); ++v) massDiag[v] = d[v];
}
// [avgEdgeLen, matSpaceBBoxSize2, avgNodeMass(3)]
void ipcref_mesh_scalars(const ipcref_mesh* h, double* out3)
{
    out3[0] = h->m->avgEdgeLen;
    out3[1] = h->m->matSpaceBBoxSize2(3);
    out3[2] = h->m->avgNodeMass(3);
}
void ipcref_mesh_surface_counts(const ipcref_mesh* h, int* n2)
{
    n2[0] = (int)h->m->SVI.size();
    n2[1] = (int)h->m->SFEdges.size();
}
void ipcref_mesh_get_surface(const ipcref_mesh* h, int* SVI, int* SFEdges2)
{
    for (int i = 0; i < h->m->SVI.size(); ++i) SVI[i] = h->m->SVI[i];
    for (size_t e = 0; e < h->m->SFEdges.size(); ++e) {
        SFEdges2[2 * e] = h->m->SFEdges[e].first;
        SFEdges2[2 * e + 1] = h->m->SFEdges[e].second;
    }
}

// ---- elasticity: Energy<3>::computeEnergyVal / computeGradient / computeHessian (Energy.cpp:195-562) ----------------------
double ipcref_elastic_energy(ipcref_mesh* h, int type, double coef)
{
    double E = 0.0;
    h->energy[type]->computeEnergyVal(*h->m, 1, h->svd, h->F, coef, E);
    return E;
}
void ipcref_elastic_gradient(ipcref_mesh* h, int type, double coef, int projectDBC, double* g)
{
    Eigen::VectorXd grad;
    h->energy[type]->computeGradient(*h->m, true, h->svd, h->F, coef, grad, projectDBC != 0);
    for (int i = 0; i < grad.size(); ++i) g[i] = grad[i];
}
// pattern from the mesh's vNeighbor (+ extra node pairs), then setZero + computeHessian; returns nnz; ia / ja are returned 0-based
int ipcref_elastic_hessian(ipcref_mesh* h, int type, double coef, int projectSPD, int projectDBC, int nExtra, const int* extraPairs2, int* ia, int* ja,
    double* a, int cap)
{
    std::vector<std::set<int>> vN = h->m->vNeighbor;
    for (int k = 0; k < nExtra; ++k) {
        vN[(size_t)extraPairs2[2 * k]].insert(extraPairs2[2 * k + 1]);
        vN[(size_t)extraPairs2[2 * k + 1]].insert(extraPairs2[2 * k]);
    }
    h->sol.reset(LinSysSolver<Eigen::VectorXi, Eigen::VectorXd>::create(LinSysSolverType::CHOLMOD));
    h->sol->set_pattern(vN, h->m->DBCVertexIds);
    h->sol->setZero();
    h->energy[type]->computeHessian(*h->m, true, h->svd, h->F, coef, h->sol.get(), projectSPD != 0, projectDBC != 0);
    const int nnz = (int)h->sol->get_ja().size(), n = h->sol->getNumRows();
    if (nnz > cap) return -nnz;
    for (int i = 0; i <= n; ++i) ia[i] = h->sol->get_ia()[i] - 1;
    for (int k = 0; k < nnz; ++k) {
        ja[k] = h->sol->get_ja()[k] - 1;
        a[k] = h->sol->get_a()[k];
    }
    return nnz;
}
// Energy::filterStepSize (Energy.cpp, get_feasible_steps.cpp:75-172): in / out step size
double ipcref_filter_step_size(ipcref_mesh* h, int type, const double* p, double stepSize)
{
    Eigen::VectorXd sd(h->m->V.rows() * 3);
    for (int i = 0; i < sd.size(); ++i) sd[i] = p[i];
    h->energy[type]->filterStepSize(*h->m, sd, stepSize);
    return stepSize;
}
// the 3x3 SVD the energies use (AutoFlipSVD over ImplicitQRSVD.h): F row-major in, U, sigma, V row-major out
void ipcref_svd3(const double* F9, double* U9, double* s3, double* V9)
{
    Eigen::Matrix3d F;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) F(i, j) = F9[3 * i + j];
    AutoFlipSVD<Eigen::Matrix3d> svd(F, Eigen::ComputeFullU | Eigen::ComputeFullV);
    for (int i = 0; i < 3; ++i) {
        s3[i] = svd.singularValues()[i];
        for (int j = 0; j < 3; ++j) {
            U9[3 * i + j] = svd.matrixU()(i, j);
            V9[3 * i + j] = svd.matrixV()(i, j);
        }
    }
}
// IglUtils::makePD on an n x n block (n = 6, 9, 12), row-major in / out
void ipcref_make_pd(int n, double* A)
{
    Eigen::MatrixXd M = toMat(A, n, n);
    IglUtils::makePD(M);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) A[n * i + j] = M(i, j);
}

// ---- stencil level: MeshCollisionUtils.hpp, BarrierFunctions.hpp ---------------------------------------------------------------
// kind 0 PP, 1 PE, 2 PT, 3 EE; X row-major 4 x 3 (unused rows ignored); g, H over 3 * nodes (row-major)
void ipcref_stencil_distance(int kind, const double* X12, double* d, double* g, double* H)
{
    Eigen::RowVector3d v[4];
    for (int k = 0; k < 4; ++k) v[k] << X12[3 * k], X12[3 * k + 1], X12[3 * k + 2];
    if (kind == 0) {
        d_PP(v[0], v[1], *d);
        if (g) {
            Eigen::Matrix<double, 6, 1> gg;
            g_PP(v[0], v[1], gg);
            for (int i = 0; i < 6; ++i) g[i] = gg[i];
        }
        if (H) {
            Eigen::Matrix<double, 6, 6> HH;
            H_PP(HH);
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j) H[6 * i + j] = HH(i, j);
        }
    }
    else if (kind == 1) {
        d_PE(v[0], v[1], v[2], *d);
        if (g) {
            Eigen::Matrix<double, 9, 1> gg;
            g_PE(v[0], v[1], v[2], gg);
            for (int i = 0; i < 9; ++i) g[i] = gg[i];
        }
        if (H) {
            Eigen::Matrix<double, 9, 9> HH;
            H_PE(v[0], v[1], v[2], HH);
            for (int i = 0; i < 9; ++i)
                for (int j = 0; j < 9; ++j) H[9 * i + j] = HH(i, j);
        }
    }
    else {
        Eigen::Matrix<double, 12, 1> gg;
        Eigen::Matrix<double, 12, 12> HH;
        if (kind == 2) {
            d_PT(v[0], v[1], v[2], v[3], *d);
            if (g) g_PT(v[0], v[1], v[2], v[3], gg);
            if (H) H_PT(v[0], v[1], v[2], v[3], HH);
        }
        else {
            d_EE(v[0], v[1], v[2], v[3], *d);
            if (g) g_EE(v[0], v[1], v[2], v[3], gg);
            if (H) H_EE(v[0], v[1], v[2], v[3], HH);
        }
        if (g)
            for (int i = 0; i < 12; ++i) g[i] = gg[i];
        if (H)
            for (int i = 0; i < 12; ++i)
                for (int j = 0; j < 12; ++j) H[12 * i + j] = HH(i, j);
    }
}
int ipcref_dtype_pt(const double* X12)
{
    Eigen::RowVector3d v[4];
    for (int k = 0; k < 4; ++k) v[k] << X12[3 * k], X12[3 * k + 1], X12[3 * k + 2];
    return dType_PT(v[0], v[1], v[2], v[3]);
}
int ipcref_dtype_ee(const double* X12)
{
    Eigen::RowVector3d v[4];
    for (int k = 0; k < 4; ++k) v[k] << X12[3 * k], X12[3 * k + 1], X12[3 * k + 2];
    return dType_EE(v[0], v[1], v[2], v[3]);
}
// classified distances (computePointTriD / computeEdgeEdgeD, MeshCollisionUtils.hpp:2279-2383)
double ipcref_classified_distance(int kind, const double* X12)
{
    Eigen::RowVector3d v[4];
    for (int k = 0; k < 4; ++k) v[k] << X12[3 * k], X12[3 * k + 1], X12[3 * k + 2];
    double d = 0.0;
    if (kind == 2) computePointTriD(v[0], v[1], v[2], v[3], d);
    else computeEdgeEdgeD(v[0], v[1], v[2], v[3], d);
    return d;
}
void ipcref_barrier(double d, double dHat, double* b, double* g, double* H)
{
    compute_b(d, dHat, *b);
    compute_g_b(d, dHat, *g);
    compute_H_b(d, dHat, *H);
}
// EE cross-product norm c, mollifier e(c, eps_x) and their derivatives (MeshCollisionUtils.hpp:2409-2912)
void ipcref_ee_mollifier(const double* X12, double eps_x, double* c, double* cg12, double* cH144, double* e, double* eg12, double* eH144)
{
    Eigen::RowVector3d v[4];
    for (int k = 0; k < 4; ++k) v[k] << X12[3 * k], X12[3 * k + 1], X12[3 * k + 2];
    computeEECrossSqNorm(v[0], v[1], v[2], v[3], *c);
    Eigen::Matrix<double, 12, 1> g;
    Eigen::Matrix<double, 12, 12> H;
    computeEECrossSqNormGradient(v[0], v[1], v[2], v[3], g);
    computeEECrossSqNormHessian(v[0], v[1], v[2], v[3], H);
    for (int i = 0; i < 12; ++i) {
        cg12[i] = g[i];
        for (int j = 0; j < 12; ++j) cH144[12 * i + j] = H(i, j);
    }
    compute_e(v[0], v[1], v[2], v[3], eps_x, *e);
    compute_e_g(v[0], v[1], v[2], v[3], eps_x, g);
    compute_e_H(v[0], v[1], v[2], v[3], eps_x, H);
    for (int i = 0; i < 12; ++i) {
        eg12[i] = g[i];
        for (int j = 0; j < 12; ++j) eH144[12 * i + j] = H(i, j);
    }
}
// IglUtils::segTriIntersect (IglUtils.hpp:214-265), the non-predicate branch of the default build
int ipcref_seg_tri_intersect(const double* X15)
{
    Eigen::RowVector3d v[5];
    for (int k = 0; k < 5; ++k) v[k] << X15[3 * k], X15[3 * k + 1], X15[3 * k + 2];
    return IglUtils::segTriIntersect(v[0], v[1], v[2], v[3], v[4]) ? 1 : 0;
}

// ---- mesh level: SelfCollisionHandler.cpp ---------------------------------------------------------------------------------------
// SelfCollisionHandler::computeConstraintSet with the spatial hash the Optimizer builds for it (Optimizer.cpp:2440-2467);
// returns counts [nActive, nPara, nPTEE]
void ipcref_constraint_set(ipcref_mesh* h, double dHat, int* n3)
{
    SpatialHash<3> sh;
    sh.build(*h->m, h->m->avgEdgeLen / 3.0);
    h->active.clear();
    h->para.clear();
    h->paraEIEJ.clear();
    h->csPTEE.clear();
    SelfCollisionHandler<3>::computeConstraintSet(*h->m, sh, dHat, h->active, h->para, h->paraEIEJ, true, h->csPTEE);
    n3[0] = (int)h->active.size();
    n3[1] = (int)h->para.size();
    n3[2] = (int)h->csPTEE.size();
}
void ipcref_constraint_get(const ipcref_mesh* h, int* active4, int* para4, int* paraEIEJ2, int* csPTEE2)
{
    for (size_t i = 0; i < h->active.size(); ++i)
        for (int k = 0; k < 4; ++k) active4[4 * i + k] = h->active[i][k];
    for (size_t i = 0; i < h->para.size(); ++i) {
        for (int k = 0; k < 4; ++k) para4[4 * i + k] = h->para[i][k];
        paraEIEJ2[2 * i] = h->paraEIEJ[i].first;
        paraEIEJ2[2 * i + 1] = h->paraEIEJ[i].second;
    }
    for (size_t i = 0; i < h->csPTEE.size(); ++i) {
        csPTEE2[2 * i] = h->csPTEE[i].first;
        csPTEE2[2 * i + 1] = h->csPTEE[i].second;
    }
}
void ipcref_constraint_put(ipcref_mesh* h, int nActive, const int* active4, int nPara, const int* para4, const int* paraEIEJ2)
{
    h->active.clear();
    h->para.clear();
    h->paraEIEJ.clear();
    for (int i = 0; i < nActive; ++i) h->active.emplace_back(active4[4 * i], active4[4 * i + 1], active4[4 * i + 2], active4[4 * i + 3]);
    for (int i = 0; i < nPara; ++i) {
        h->para.emplace_back(para4[4 * i], para4[4 * i + 1], para4[4 * i + 2], para4[4 * i + 3]);
        h->paraEIEJ.emplace_back(paraEIEJ2[2 * i], paraEIEJ2[2 * i + 1]);
    }
}
// the barrier part of the incremental potential as Optimizer::computeEnergyVal sums it (Optimizer.cpp:3262-3300):
// kappa * (sum_i b(d_i) over the active set, multiplicities inside evaluateConstraints' coefficients) + mollified pairs
double ipcref_barrier_energy(ipcref_mesh* h, double dHat, double kappa)
{
    Eigen::VectorXd val;
    SelfCollisionHandler<3>::evaluateConstraints(*h->m, h->active, val);
    double E = 0.0;
    for (int i = 0; i < val.size(); ++i) {
        double b;
        compute_b(val[i], dHat, b);
        // MMCVID[3] < -1 encodes the multiplicity of a merged PP / PE pair (Optimizer.cpp:3307-3311)
        const int duplication = h->active[(size_t)i][3];
        E += (duplication < -1 ? -duplication : 1) * b;
    }
    // mollified (nearly parallel edge-edge) pairs, the glue of Optimizer.cpp:3316-3349: b(d) * e(c, eps_x)
    Eigen::VectorXd pv;
    SelfCollisionHandler<3>::evaluateConstraints(*h->m, h->para, pv);
    for (int i = 0; i < pv.size(); ++i) {
        const MMCVID& c = h->para[(size_t)i];
        double eps_x, e, b;
        if (c[3] >= 0) {
            compute_eps_x(*h->m, c[0], c[1], c[2], c[3], eps_x);
            compute_e(h->m->V.row(c[0]), h->m->V.row(c[1]), h->m->V.row(c[2]), h->m->V.row(c[3]), eps_x, e);
        }
        else {
            const std::pair<int, int>& eI = h->m->SFEdges[(size_t)h->paraEIEJ[(size_t)i].first];
            const std::pair<int, int>& eJ = h->m->SFEdges[(size_t)h->paraEIEJ[(size_t)i].second];
            compute_eps_x(*h->m, eI.first, eI.second, eJ.first, eJ.second, eps_x);
            compute_e(h->m->V.row(eI.first), h->m->V.row(eI.second), h->m->V.row(eJ.first), h->m->V.row(eJ.second), eps_x, e);
        }
        compute_b(pv[i], dHat, b);
        E += b * e;
    }
    return kappa * E;
}
void ipcref_barrier_gradient(ipcref_mesh* h, double dHat, double kappa, double* g)
{
    // Optimizer::computeGradient (Optimizer.cpp:3455-3490): g += kappa * J^T g_b(d), then the mollified pairs
    Eigen::VectorXd val, grad = Eigen::VectorXd::Zero(h->m->V.rows() * 3);
    SelfCollisionHandler<3>::evaluateConstraints(*h->m, h->active, val);
    for (int i = 0; i < val.size(); ++i) compute_g_b(val[i], dHat, val[i]);
    SelfCollisionHandler<3>::leftMultiplyConstraintJacobianT(*h->m, h->active, val, grad, kappa);
    SelfCollisionHandler<3>::augmentParaEEGradient(*h->m, h->para, h->paraEIEJ, grad, dHat, kappa);
    for (int i = 0; i < grad.size(); ++i) g[i] = grad[i];
}
// pattern = mesh connectivity + augmentConnectivity of both sets; values = augmentIPHessian + augmentParaEEHessian only
int ipcref_barrier_hessian(ipcref_mesh* h, double dHat, double kappa, int projectDBC, int* ia, int* ja, double* a, int cap)
{
    std::vector<std::set<int>> vN = h->m->vNeighbor;
    SelfCollisionHandler<3>::augmentConnectivity(*h->m, h->active, vN);
    SelfCollisionHandler<3>::augmentConnectivity(*h->m, h->para, h->paraEIEJ, vN);
    h->sol.reset(LinSysSolver<Eigen::VectorXi, Eigen::VectorXd>::create(LinSysSolverType::CHOLMOD));
    h->sol->set_pattern(vN, h->m->DBCVertexIds);
    h->sol->setZero();
    SelfCollisionHandler<3>::augmentIPHessian(*h->m, h->active, h->sol.get(), dHat, kappa, projectDBC != 0);
    SelfCollisionHandler<3>::augmentParaEEHessian(*h->m, h->para, h->paraEIEJ, h->sol.get(), dHat, kappa, projectDBC != 0);
    const int nnz = (int)h->sol->get_ja().size(), n = h->sol->getNumRows();
    if (nnz > cap) return -nnz;
    for (int i = 0; i <= n; ++i) ia[i] = h->sol->get_ia()[i] - 1;
    for (int k = 0; k < nnz; ++k) {
        ja[k] = h->sol->get_ja()[k] - 1;
        a[k] = h->sol->get_a()[k];
    }
    return nnz;
}
// full CCD sweep (largestFeasibleStepSize_CCD) with the hash the Optimizer builds for it (Optimizer.cpp:1963-1996)
double ipcref_full_ccd(ipcref_mesh* h, const double* p, double slackness, double stepSize)
{
    Eigen::VectorXd sd(h->m->V.rows() * 3);
    for (int i = 0; i < sd.size(); ++i) sd[i] = p[i];
    SpatialHash<3> sh;
    sh.build(*h->m, sd, stepSize, h->m->avgEdgeLen / 3.0);
    std::vector<std::pair<int, int>> cand;
    SelfCollisionHandler<3>::largestFeasibleStepSize_CCD(*h->m, sh, sd, slackness, cand, stepSize);
    return stepSize;
}
// CCD restricted to the PT / EE candidate list of the last constraint-set build (largestFeasibleStepSize, CFL path)
double ipcref_partial_ccd(ipcref_mesh* h, const double* p, double slackness, double stepSize)
{
    Eigen::VectorXd sd(h->m->V.rows() * 3);
    for (int i = 0; i < sd.size(); ++i) sd[i] = p[i];
    SpatialHash<3> sh;
    std::vector<std::pair<int, int>> cand;
    SelfCollisionHandler<3>::largestFeasibleStepSize(*h->m, sh, sd, slackness, h->csPTEE, cand, stepSize);
    return stepSize;
}
int ipcref_is_intersected(ipcref_mesh* h)
{
    SpatialHash<3> sh;
    sh.build(*h->m, h->m->avgEdgeLen / 3.0);
    return SelfCollisionHandler<3>::checkEdgeTriIntersectionIfAny(*h->m, sh) ? 0 : 1; // the reference returns true when clean
}


// ---- half-space (HalfSpace.cpp, CollisionObject.h:323-401): active vertices, barrier energy / gradient / Hessian, ray step bound ----
// out2 = [nActive, energy]; grad over 3 nV; Hessian into the mesh pattern (CSR, 0-based); active (capacity nV) receives the vertex ids
int ipcref_halfspace_eval(ipcref_mesh* h, const double* origin3, const double* normal3, double dHat, double kappa, int projectDBC, int* active,
    double* energy, double* grad, int* ia, int* ja, double* a, int cap)
{
    Eigen::Vector3d o(origin3[0], origin3[1], origin3[2]), n(normal3[0], normal3[1], normal3[2]), v0(0.0, 0.0, 0.0);
    HalfSpace<3> hs(o, n, v0, 0.0);
    std::vector<int> act;
    hs.computeConstraintSet(*h->m, dHat, act);
    for (size_t i = 0; i < act.size(); ++i) active[i] = act[i];
    Eigen::VectorXd val;
    hs.evaluateConstraints(*h->m, act, val);
    double E = 0.0;
    for (int i = 0; i < val.size(); ++i) {
        double b;
        compute_b(val[i], dHat, b);
        E += b;
    }
    *energy = kappa * E;
    Eigen::VectorXd g = Eigen::VectorXd::Zero(h->m->V.rows() * 3);
    for (int i = 0; i < val.size(); ++i) compute_g_b(val[i], dHat, val[i]);
    hs.leftMultiplyConstraintJacobianT(*h->m, act, val, g, kappa);
    for (int i = 0; i < g.size(); ++i) grad[i] = g[i];
    h->sol.reset(LinSysSolver<Eigen::VectorXi, Eigen::VectorXd>::create(LinSysSolverType::CHOLMOD));
    h->sol->set_pattern(h->m->vNeighbor, h->m->DBCVertexIds);
    h->sol->setZero();
    hs.augmentIPHessian(*h->m, act, h->sol.get(), dHat, kappa, projectDBC != 0);
    const int nnz = (int)h->sol->get_ja().size(), nr = h->sol->getNumRows();
    if (nnz > cap) return -nnz;
    for (int i = 0; i <= nr; ++i) ia[i] = h->sol->get_ia()[i] - 1;
    for (int k = 0; k < nnz; ++k) {
        ja[k] = h->sol->get_ja()[k] - 1;
        a[k] = h->sol->get_a()[k];
    }
    return (int)act.size();
}
double ipcref_halfspace_step_bound(ipcref_mesh* h, const double* origin3, const double* normal3, const double* p, double slackness, double stepSize)
{
    Eigen::Vector3d o(origin3[0], origin3[1], origin3[2]), n(normal3[0], normal3[1], normal3[2]), v0(0.0, 0.0, 0.0);
    HalfSpace<3> hs(o, n, v0, 0.0);
    Eigen::VectorXd sd(h->m->V.rows() * 3);
    for (int i = 0; i < sd.size(); ++i) sd[i] = p[i];
    std::vector<int> AHat;
    hs.largestFeasibleStepSize(*h->m, sd, slackness, AHat, stepSize);
    return stepSize;
}

// HalfSpace::move (HalfSpace.cpp:389-416): the origin after the move; returns the fraction that is left
double ipcref_halfspace_move(ipcref_mesh* h, const double* origin3, const double* normal3, const double* delta3, double slackness, double* originOut)
{
    Eigen::Vector3d o(origin3[0], origin3[1], origin3[2]), n(normal3[0], normal3[1], normal3[2]), v0(0.0, 0.0, 0.0), d(delta3[0], delta3[1], delta3[2]);
    HalfSpace<3> hs(o, n, v0, 0.0);
    SpatialHash<3> sh;
    double left = 0.0;
    hs.move(d, *h->m, sh, slackness, left);
    for (int c = 0; c < 3; ++c) originOut[c] = hs.origin[c];
    return left;
}

} // extern "C"
