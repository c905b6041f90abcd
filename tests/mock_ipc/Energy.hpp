// TEST INFRASTRUCTURE -- stand-in for the reference's src/Energy/Energy.hpp: the virtual interface of Energy<dim>
// (Energy.hpp:26-59, 129-131) with empty default bodies, so that include/adapters/HipElasticEnergy.hpp compiles here.
#pragma once
#include "LinSysSolver.hpp"
#include "Mesh.hpp"
#include <vector>

namespace IPC {

template <class MatrixType>
class AutoFlipSVD { // src/Utils/AutoFlipSVD.hpp: only named in the signatures below
};

template <int dim>
class Energy {
protected:
    const bool needElemInvSafeGuard; // Energy.hpp:30

public:
    explicit Energy(bool p_needElemInvSafeGuard) : needElemInvSafeGuard(p_needElemInvSafeGuard) {} // :33
    virtual ~Energy(void) {}
    bool getNeedElemInvSafeGuard(void) const { return needElemInvSafeGuard; } // :37

    virtual void computeEnergyVal(const Mesh<dim>&, int, std::vector<AutoFlipSVD<Eigen::Matrix<double, dim, dim>>>&,
        std::vector<Eigen::Matrix<double, dim, dim>>&, double, double&) const {} // :41-45
    virtual void computeGradient(const Mesh<dim>&, bool, std::vector<AutoFlipSVD<Eigen::Matrix<double, dim, dim>>>&,
        std::vector<Eigen::Matrix<double, dim, dim>>&, double, Eigen::VectorXd&, bool = true) const {} // :46-51
    virtual void computeHessian(const Mesh<dim>&, bool, std::vector<AutoFlipSVD<Eigen::Matrix<double, dim, dim>>>&,
        std::vector<Eigen::Matrix<double, dim, dim>>&, double, LinSysSolver<Eigen::VectorXi, Eigen::VectorXd>*, bool = true,
        bool = true) const {} // :52-58
    virtual void filterStepSize(const Mesh<dim>&, const Eigen::VectorXd&, double&) const {} // :129-131
};

} // namespace IPC
