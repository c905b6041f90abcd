// ORACLE / TEST INFRASTRUCTURE: libigl's avg_edge_length restated (un-vendored): the mean of |V(F(i,j)) - V(F(i,(j+1)%cols))|
// over all rows i and columns j -- for tetrahedra that is four of the six edges of every element, interior edges counted once
// per element.  The reference uses the value as the cell size of its spatial hash (avgEdgeLen / 3) and for tolerances of its
// 2-D debugging helpers; the constraint sets and step bounds it computes do not depend on it.
#pragma once
#include <Eigen/Core>
namespace igl {
template <class DV, class DF>
inline double avg_edge_length(const Eigen::MatrixBase<DV>& V, const Eigen::MatrixBase<DF>& F)
{
    double avg = 0.0;
    long count = 0;
    for (Eigen::Index i = 0; i < F.rows(); ++i)
        for (Eigen::Index j = 0; j < F.cols(); ++j) {
            ++count;
            avg += (V.row(F(i, j)) - V.row(F(i, (j + 1) % F.cols()))).norm();
        }
    return count ? avg / (double)count : 0.0;
}
} // namespace igl
