// HipElasticEnergy -- Energy<3> subclass that evaluates the elasticity term of ipc-sim/IPC on an MI355X: replaces
// NeoHookeanEnergy<3> / FixedCoRotEnergy<3> created at src/main.cpp:1386 (Energy.hpp:26-131).
//
// It shares one ipcgpu_ctx with the other adapters (the Optimizer adapter owns it).  computeHessian adds into the solver
// it is handed (Energy.hpp:52-58):
//   * a HipLinSysSolver on the same ctx: the element Hessians are added to its values in HBM, nothing crosses PCIe;
//   * any other LinSysSolver (CHOLMOD, AMGCL, Eigen -- the A/B run inside one binary): the projected Hessian is
//     assembled on the device in the ctx's own CSR (same pattern: LinSysSolver::set_pattern is bit-identical), read back
//     once and added to the solver's own array through get_a().
// The mesh contract goes over once with hipUploadMesh(): V_rest, F and the per-element arrays Mesh<3> already holds.
#pragma once
#include "Energy.hpp"
#include "HipLinSysSolver.hpp"
#include <ipcgpu.h>
#include <stdexcept>
#include <vector>

namespace IPC {

// Mesh<3> -> device: rest shape, elements, the reference's own restTriInv / triArea / lumped mass / Lame arrays
// (Mesh.hpp:147-164) and the Dirichlet types (Mesh.hpp:131-144).  E / nu / rho are still passed: the library derives
// quantities from them that Mesh<3> does not store (average edge length, bounding box).
inline void hipUploadMesh(ipcgpu_ctx* ctx, const Mesh<3>& data, double YM, double PR, double density)
{
    auto chk = [](int rc) {
        if (rc < 0) throw std::runtime_error(ipcgpu_last_error());
    };
    const int nV = (int)data.V_rest.rows(), nT = (int)data.F.rows();
    chk(ipcgpu_set_mesh(ctx, nV, nT, data.V_rest.data(), data.F.data(), YM, PR, density));
    std::vector<double> A(9 * (size_t)nT), mass((size_t)nV);
    for (int t = 0; t < nT; ++t)
        for (int k = 0; k < 9; ++k) A[9 * (size_t)t + k] = data.restTriInv[t].data()[k];
    for (int v = 0; v < nV; ++v) mass[v] = data.massMatrix.coeff(v, v);
    chk(ipcgpu_set_mesh_features(ctx, A.data(), data.triArea.data(), mass.data(), data.u.data(), data.lambda.data()));
    chk(ipcgpu_clear_dbc(ctx));
    for (int type = 1; type <= 2; ++type) {
        std::vector<int> ids;
        for (int v = 0; v < nV; ++v)
            if ((int)data.vertexDBCType[v] == type) ids.push_back(v);
        if (!ids.empty()) chk(ipcgpu_set_dbc(ctx, (int)ids.size(), ids.data(), type));
    }
}

class HipElasticEnergy : public Energy<3> {
    typedef LinSysSolver<Eigen::VectorXi, Eigen::VectorXd> Solver;
    typedef HipLinSysSolver<Eigen::VectorXi, Eigen::VectorXd> HipSolver;
    ipcgpu_ctx* ctx;
    mutable std::vector<double> scratch_;
    mutable std::vector<int> ia0_, ja0_;

    static void chk(int rc)
    {
        if (rc < 0) throw std::runtime_error(ipcgpu_last_error());
    }
    int energyType_;
    mutable bool typeSet_ = false;
    void positions(const Mesh<3>& data) const
    {
        if (!typeSet_) { // the type is a property of the mesh on the device: handed over once the mesh is there (hipUploadMesh)
            chk(ipcgpu_set_energy_type(ctx, energyType_));
            typeSet_ = true;
        }
        chk(ipcgpu_set_positions(ctx, data.V.data()));
    }

public:
    // energyType: 0 = neo-Hookean (needs the inversion safeguard), 1 = fixed corotated (Config.cpp:107-111)
    // may be constructed before the mesh is uploaded (the optimizer adapter creates it ahead of its base class)
    HipElasticEnergy(ipcgpu_ctx* shared, int energyType = 0) : Energy<3>(energyType == 0), ctx(shared), energyType_(energyType) {}

    void computeEnergyVal(const Mesh<3>& data, int, std::vector<AutoFlipSVD<Eigen::Matrix<double, 3, 3>>>&,
        std::vector<Eigen::Matrix<double, 3, 3>>&, double coef, double& energyVal) const override
    {
        positions(data);
        chk(ipcgpu_elastic_energy(ctx, coef, &energyVal));
    }
    void computeGradient(const Mesh<3>& data, bool, std::vector<AutoFlipSVD<Eigen::Matrix<double, 3, 3>>>&,
        std::vector<Eigen::Matrix<double, 3, 3>>&, double coef, Eigen::VectorXd& gradient, bool projectDBC = true) const override
    {
        gradient.resize(data.V.rows() * 3);
        positions(data);
        chk(ipcgpu_elastic_gradient(ctx, coef, projectDBC, gradient.data()));
    }
    void computeHessian(const Mesh<3>& data, bool, std::vector<AutoFlipSVD<Eigen::Matrix<double, 3, 3>>>&,
        std::vector<Eigen::Matrix<double, 3, 3>>&, double coef, Solver* linSysSolver, bool /*projectSPD: always on the Newton path,
        Optimizer.cpp:3622*/ = true, bool projectDBC = true) const override
    {
        positions(data);
        HipSolver* hip = dynamic_cast<HipSolver*>(linSysSolver);
        if (hip && hip->context() == ctx) {
            hip->flush(); // pending host-side sets come first, as they would on the host
            chk(ipcgpu_elastic_hessian_add(ctx, coef, projectDBC));
            return;
        }
        // a host solver: assemble in the context's own CSR (same pattern), one read-back, add into the solver's array
        const Eigen::VectorXi& ia = linSysSolver->get_ia();
        const Eigen::VectorXi& ja = linSysSolver->get_ja();
        const int n = linSysSolver->getNumRows(), base = ia[0]; // 1-based until a solver subclass rebased them
        ia0_.resize((size_t)n + 1);
        ja0_.resize((size_t)ja.size());
        for (int i = 0; i <= n; ++i) ia0_[i] = ia[i] - base;
        for (long k = 0; k < ja.size(); ++k) ja0_[(size_t)k] = ja[k] - base;
        hipSetPattern(ctx, n, ia0_.data(), ja0_.data());
        chk(ipcgpu_linsys_set_zero(ctx));
        chk(ipcgpu_elastic_hessian_add(ctx, coef, projectDBC));
        scratch_.resize((size_t)ja.size());
        chk(ipcgpu_linsys_get_values(ctx, scratch_.data()));
        Eigen::VectorXd& a = linSysSolver->get_a();
        for (long k = 0; k < a.size(); ++k) a[k] += scratch_[(size_t)k];
    }
    void filterStepSize(const Mesh<3>& data, const Eigen::VectorXd& searchDir, double& stepSize) const override
    {
        positions(data);
        chk(ipcgpu_filter_step_size(ctx, searchDir.data(), &stepSize));
    }
};

} // namespace IPC
