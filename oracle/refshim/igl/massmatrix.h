// ORACLE / TEST INFRASTRUCTURE: Mesh.cpp carries its own copy of libigl's massmatrix body (Mesh.cpp:120-412) and
// calls three small libigl helpers from it.  They are restated here from libigl's documented behaviour:
// doublearea from edge lengths (Kahan's stable Heron formula on the sorted lengths), repmat, sparse (sum of
// triplets).  Only codimension-2 components (triangle meshes under `shapes`) reach doublearea / repmat; the
// tetrahedral path uses sparse alone.
#pragma once
#include <Eigen/Sparse>
#include <algorithm>
#include <cmath>
#include <vector>
namespace igl {
enum MassMatrixType { MASSMATRIX_TYPE_BARYCENTRIC = 0,
    MASSMATRIX_TYPE_VORONOI = 1,
    MASSMATRIX_TYPE_FULL = 2,
    MASSMATRIX_TYPE_DEFAULT = 3,
    NUM_MASSMATRIX_TYPE = 4 };
template <class L, class T, class D>
inline void doublearea(const Eigen::MatrixBase<L>& ul, T nan_replacement, Eigen::MatrixBase<D>& dblA)
{
    const Eigen::Index m = ul.rows();
    dblA.derived().resize(m, 1);
    for (Eigen::Index i = 0; i < m; ++i) {
        double l[3] = { (double)ul(i, 0), (double)ul(i, 1), (double)ul(i, 2) };
        std::sort(l, l + 3, [](double a, double b) { return a > b; });
        const double arg = (l[0] + (l[1] + l[2])) * (l[2] - (l[0] - l[1])) * (l[2] + (l[0] - l[1])) * (l[0] + (l[1] - l[2]));
        double v = 2.0 * 0.25 * std::sqrt(arg);
        if (v != v) v = (double)nan_replacement;
        dblA(i) = v;
    }
}
template <class A, class B>
inline void repmat(const Eigen::MatrixBase<A>& a, int r, int c, Eigen::MatrixBase<B>& b)
{
    b.derived().resize(r * a.rows(), c * a.cols());
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j)
            for (Eigen::Index jj = 0; jj < a.cols(); ++jj)
                for (Eigen::Index ii = 0; ii < a.rows(); ++ii) b(i * a.rows() + ii, j * a.cols() + jj) = a(ii, jj);
}
template <class I, class J, class V, class T>
inline void sparse(const Eigen::MatrixBase<I>& i, const Eigen::MatrixBase<J>& j, const Eigen::MatrixBase<V>& v, size_t m, size_t n, Eigen::SparseMatrix<T>& x)
{
    std::vector<Eigen::Triplet<T>> t;
    t.reserve((size_t)i.size());
    for (Eigen::Index k = 0; k < i.size(); ++k) t.emplace_back((int)i(k), (int)j(k), (T)v(k));
    x.resize((Eigen::Index)m, (Eigen::Index)n);
    x.setFromTriplets(t.begin(), t.end());
}
} // namespace igl
