// HipHalfSpace: analytic half-space obstacle of the hot path with its vertex constraint set in HBM.
//   HalfSpace::init                              src/CollisionObject/HalfSpace.cpp:41-85
//   CollisionObject::computeConstraintSet        src/CollisionObject/CollisionObject.h:323-351   -> build
//   evaluateConstraint + barrier energy          HalfSpace.cpp:106-111, Optimizer.cpp:3254-3267  -> energy
//   leftMultiplyConstraintJacobianT              HalfSpace.cpp:121-143                           -> gradientAdd
//   augmentIPHessian                             HalfSpace.cpp:169-214                           -> hessianAdd
//   largestFeasibleStepSize                      HalfSpace.cpp:242-269                           -> stepBound
//   CollisionObject::isIntersected               CollisionObject.h:386-401                       -> intersected
#pragma once
#include "common.h"
#include <vector>

namespace ipcgpu {

class HipHalfSpace {
public:
    HipHalfSpace(hipStream_t s, const double* origin3, const double* normal3);
    hipStream_t stream;
    double n[3], D, origin[3];
    std::vector<int> set; // activeSet[coI]: vertex ids, ascending surface-vertex order
    DevBuf<int> d_set;

    int build(int nSVI, const int* svi_dev, const double* x_dev, const int* dbc_dev, double dHat);
    void setSet(int n, const int* verts);
    double energy(const double* x_dev, double dHat, double kappa);
    void gradientAdd(const double* x_dev, double dHat, double kappa, double* grad_dev);
    void hessianAdd(const double* x_dev, const int* dbc_dev, const int* rowBase_dev, const int* rowLen_dev, double dHat, double kappa, int projectDBC,
        double* a_dev);
    double stepBound(int nSVI, const int* svi_dev, const double* x_dev, const int* dbc_dev, const double* p_dev, double slackness, double stepSize);
    bool intersected(int nV, const double* x_dev, const int* dbc_dev);
    // HalfSpace::move (HalfSpace.cpp:389-416): the plane is displaced by the largest fraction <= 1 of delta that keeps `slackness` of every surface
    // node's distance (Dirichlet nodes included); returns the fraction that is left
    double move(int nSVI, const int* svi_dev, const double* x_dev, const double* delta3, double slackness);
    void evalDist2(const std::vector<int>& verts, const double* x_dev, std::vector<double>& d2);
    // lagged friction (HalfSpace.cpp:272-381, C0 clamping): activeSet_lastH / lambda_lastH of this plane
    double friction = 0.0; // Base::friction
    std::vector<int> lagSet;
    DevBuf<int> d_lagSet;
    DevBuf<double> d_lagLambda;
    void lagClear() { lagSet.clear(); }
    void lagUpdate(const double* x_dev, double dHat, double kappa); // Optimizer.cpp:1560-1573 on the current `set`
    double frictionEnergy(const double* x_dev, const double* xt_dev, double eps2);
    void frictionGradientAdd(const double* x_dev, const double* xt_dev, double eps2, double* grad_dev);
    void frictionHessianAdd(const double* x_dev, const double* xt_dev, const int* dbc_dev, const int* rowBase_dev, const int* rowLen_dev, double eps2,
        int projectDBC, double* a_dev);

private:
    DevBuf<int> flags_, count_, ids_;
    DevBuf<char> tmp_;
    DevBuf<double> partial_, vals_;
    DevBuf<unsigned long long> minOut_;
};

} // namespace ipcgpu
