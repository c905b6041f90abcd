// TEST INFRASTRUCTURE -- Eigen-free stand-in for the reference's src/LinSysSolver/LinSysSolver.hpp (same class, same
// protected members, same virtual signatures, file:line cited) so that include/adapters/HipLinSysSolver.hpp can be compiled
// and exercised here.  Bodies are restated from the documented behaviour, not copied: with the real header on the include
// path instead of this directory the adapter compiles against the reference unchanged.
#pragma once
#include "Types.hpp"
#include <Eigen/Eigen>
#include <Eigen/Sparse>
#include <cassert>
#include <limits>
#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

namespace IPC {

enum class LinSysSolverType { // LinSysSolver.hpp:25-29, plus the value the adapter adds
    CHOLMOD,
    AMGCL,
    EIGEN,
    HIP
};

template <typename vectorTypeI, typename vectorTypeS>
class LinSysSolver {
protected: // LinSysSolver.hpp:33-37
    int numRows = 0;
    Eigen::VectorXi ia, ja;
    std::vector<std::map<int, int>> IJ2aI;
    Eigen::VectorXd a;

public:
    virtual ~LinSysSolver(void) {}
    virtual LinSysSolverType type() const = 0;

    // LinSysSolver.hpp:46-150: symmetric-upper CSR with 3 x 3 node blocks; ia / ja 1-based, IJ2aI[row][col] = 0-based slot.
    // Row 3 v + r: the diagonal block's columns 3 v + r .. 3 v + 2, then the three columns of every neighbour > v, ascending.
    virtual void set_pattern(const std::vector<std::set<int>>& vNeighbor, const std::set<int>& fixedVert)
    {
        (void)fixedVert; // "fixed verts nnz entries are not eliminated" (:149)
        const int nV = (int)vNeighbor.size();
        numRows = nV * DIM;
        ia.resize(numRows + 1);
        IJ2aI.assign(numRows, std::map<int, int>());
        std::vector<int> cols;
        ia[0] = 1;
        for (int v = 0; v < nV; ++v)
            for (int r = 0; r < DIM; ++r) {
                const int row = v * DIM + r;
                for (int c = r; c < DIM; ++c) {
                    IJ2aI[row][v * DIM + c] = (int)cols.size();
                    cols.push_back(v * DIM + c + 1);
                }
                for (int nb : vNeighbor[v])
                    if (nb > v)
                        for (int c = 0; c < DIM; ++c) {
                            IJ2aI[row][nb * DIM + c] = (int)cols.size();
                            cols.push_back(nb * DIM + c + 1);
                        }
                ia[row + 1] = (int)cols.size() + 1;
            }
        ja.resize((long)cols.size());
        for (size_t k = 0; k < cols.size(); ++k) ja[(long)k] = cols[k];
        a.resize(ja.size());
    }
    virtual void analyze_pattern(void) = 0; // :207
    virtual bool factorize(void) = 0; // :209
    virtual void solve(Eigen::VectorXd& rhs, Eigen::VectorXd& result) = 0; // :211-213
    virtual void multiply(const Eigen::VectorXd& x, Eigen::VectorXd& Ax) // :215-232
    {
        Ax.setZero(numRows);
        for (int rowI = 0; rowI < numRows; ++rowI)
            for (const auto& colI : IJ2aI[rowI]) {
                Ax[rowI] += a[colI.second] * x[colI.first];
                if (rowI != colI.first) Ax[colI.first] += a[colI.second] * x[rowI];
            }
    }
    virtual double coeffMtr(int rowI, int colI) const // :239-256
    {
        if (rowI > colI) std::swap(rowI, colI);
        const auto finder = IJ2aI[rowI].find(colI);
        return finder != IJ2aI[rowI].end() ? a[finder->second] : 0.0;
    }
    virtual void setCoeff(int rowI, int colI, double val) // :331-339
    {
        if (rowI <= colI) a[IJ2aI[rowI].find(colI)->second] = val;
    }
    virtual void setZero(void) { a.setZero(); } // :348-351
    virtual void setUnit_row(int rowI) // :352-359
    {
        for (const auto& colIter : IJ2aI[rowI]) a[colIter.second] = (colIter.first == rowI);
    }
    virtual void setUnit_col(int colI, const std::set<int>& rowVIs) // :370-386
    {
        for (const auto& rowVI : rowVIs)
            for (int dimI = 0; dimI < DIM; ++dimI) {
                const int rowI = rowVI * DIM + dimI;
                if (rowI <= colI) {
                    const auto finder = IJ2aI[rowI].find(colI);
                    if (finder != IJ2aI[rowI].end()) a[finder->second] = (rowI == colI);
                }
            }
    }
    virtual void addCoeff(int rowI, int colI, double val) // :402-410
    {
        if (rowI <= colI) a[IJ2aI[rowI].find(colI)->second] += val;
    }
    virtual void precondition_diag(const Eigen::VectorXd& input, Eigen::VectorXd& output) // :411-420
    {
        output.resize(numRows);
        for (int rowI = 0; rowI < numRows; ++rowI) output[rowI] = input[rowI] / a[IJ2aI[rowI].find(rowI)->second];
    }
    virtual void getMaxDiag(double& maxDiag) // :421-431
    {
        maxDiag = -std::numeric_limits<double>::infinity();
        for (int rowI = 0; rowI < numRows; ++rowI) maxDiag = std::max(maxDiag, a[IJ2aI[rowI].find(rowI)->second]);
    }
    virtual int getNumRows(void) const { return numRows; } // :451
    virtual int getNumNonzeros(void) const { return (int)a.size(); } // :455
    virtual const std::vector<std::map<int, int>>& getIJ2aI(void) const { return IJ2aI; } // :459
    virtual Eigen::VectorXi& get_ia(void) { return ia; } // :463-466
    virtual Eigen::VectorXi& get_ja(void) { return ja; }
    virtual Eigen::VectorXd& get_a(void) { return a; }
    virtual const Eigen::VectorXd& get_a(void) const { return a; }
};

} // namespace IPC
