// HipLinSysSolver -- the LinSysSolver subclass a maintainer drops into src/LinSysSolver/ to put the sparse Cholesky of
// ipc-sim/IPC on an MI355X (replaces CHOLMODSolver, src/LinSysSolver/CHOLMODSolver.cpp:18-195).
//
// Ownership: the CSR values live in HBM, inside the ipcgpu_ctx this solver was given (one ctx per Optimizer, shared with the
// energy / collision adapters, so that Energy::computeHessian(..., LinSysSolver*, ...) can add into the matrix it is handed
// without the values ever visiting the host -- Energy.hpp:52-58).  The reference's host-side entry points keep working:
//   addCoeff / setCoeff / setUnit_* (LinSysSolver.hpp:331-410, called from tbb::parallel_for bodies on distinct rows,
//   Energy.cpp:317-327, Optimizer.cpp:3638-3668) accumulate into a host-side pending update (a delta per entry plus a
//   "set" mask) that is flushed to the device in one pass before anything reads the matrix (factorize / multiply / get_a /
//   coeffMtr / a device-side add).  Per entry the order "adds, set, adds" is preserved: a set clears the pending delta of
//   that entry, later adds accumulate on top of the set value.  A set does not order against device-side adds that happen
//   while it is still pending; the adapters that add on the device flush first.
// Needs: the reference's LinSysSolver.hpp (Eigen), include/ipcgpu.h, -lipcgpu.
#pragma once
#include "LinSysSolver.hpp"
#include <ipcgpu.h>
#include <stdexcept>
#include <vector>

// the enum value a maintainer adds to LinSysSolverType (LinSysSolver.hpp:25-29); overridable so that the header also compiles
// against an unmodified tree
#ifndef IPCGPU_LINSYSSOLVER_TYPE
#define IPCGPU_LINSYSSOLVER_TYPE LinSysSolverType::HIP
#endif

namespace IPC {

// Hand a block-structured symmetric-upper CSR (0-based; what LinSysSolver::set_pattern builds) to the context.  When the
// context holds the mesh the matrix belongs to, the pattern goes in as node pairs on top of the mesh adjacency
// (ipcgpu_linsys_set_pattern: the element kernels then know their slots and Energy::computeHessian can add in HBM); the
// library rebuilds ia / ja with the rules of LinSysSolver.hpp:46-150 and the size is checked.  Otherwise (a matrix without a
// mesh, Diagnostic.cpp:367-392) the CSR is taken as it is.
inline void hipSetPattern(ipcgpu_ctx* ctx, int numRows, const int* ia0, const int* ja0)
{
    auto chk = [](int rc) {
        if (rc < 0) throw std::runtime_error(ipcgpu_last_error());
    };
    int nV = 0, nT = 0;
    chk(ipcgpu_get_mesh_dims(ctx, &nV, &nT));
    if (nV > 0 && 3 * nV == numRows) {
        std::vector<int> pairs;
        for (int v = 0; v < nV; ++v)
            for (int k = ia0[3 * v] + 3; k < ia0[3 * v + 1]; k += 3) {
                pairs.push_back(v);
                pairs.push_back(ja0[k] / 3);
            }
        chk(ipcgpu_linsys_set_pattern(ctx, (int)(pairs.size() / 2), pairs.data()));
        int rows = 0, nnz = 0;
        chk(ipcgpu_linsys_get_dims(ctx, &rows, &nnz));
        if (rows == numRows && nnz == ia0[numRows]) return;
        // the pattern does not contain the mesh adjacency (not a Hessian of this mesh): plain CSR below
    }
    chk(ipcgpu_linsys_set_pattern_csr(ctx, numRows, ia0, ja0));
}

template <typename vectorTypeI, typename vectorTypeS>
class HipLinSysSolver : public LinSysSolver<vectorTypeI, vectorTypeS> {
    typedef LinSysSolver<vectorTypeI, vectorTypeS> Base;

protected:
    ipcgpu_ctx* ctx = nullptr;
    bool ownsCtx = false;
    mutable std::vector<double> delta_, setVal_;
    mutable std::vector<unsigned char> isSet_;
    mutable bool dirty_ = false, anySet_ = false;

    static void chk(int rc)
    {
        if (rc < 0) throw std::runtime_error(ipcgpu_last_error());
    }
    int slot(int rowI, int colI) const
    {
        const auto finder = Base::IJ2aI[rowI].find(colI);
        if (finder == Base::IJ2aI[rowI].end()) throw std::out_of_range("HipLinSysSolver: entry outside the pattern");
        return finder->second;
    }

public:
    // shared == nullptr: a private context (stand-alone use as in Diagnostic.cpp:367-392)
    explicit HipLinSysSolver(ipcgpu_ctx* shared = nullptr, int device = 0) : ctx(shared)
    {
        if (!ctx) {
            chk(ipcgpu_ctx_create(device, &ctx));
            ownsCtx = true;
        }
    }
    ~HipLinSysSolver() override
    {
        if (ownsCtx) ipcgpu_ctx_destroy(ctx);
    }
    HipLinSysSolver(const HipLinSysSolver&) = delete;
    HipLinSysSolver& operator=(const HipLinSysSolver&) = delete;

    ipcgpu_ctx* context() const { return ctx; }
    LinSysSolverType type() const override { return IPCGPU_LINSYSSOLVER_TYPE; }

    // host-side addCoeff / setCoeff since the last flush -> HBM (one pass over the values)
    void flush() const
    {
        if (!dirty_) return;
        chk(ipcgpu_linsys_apply_host_updates(ctx, delta_.data(), anySet_ ? isSet_.data() : nullptr, anySet_ ? setVal_.data() : nullptr));
        std::fill(delta_.begin(), delta_.end(), 0.0);
        if (anySet_) std::fill(isSet_.begin(), isSet_.end(), (unsigned char)0);
        dirty_ = anySet_ = false;
    }
    // HBM -> the host mirror Base::a (what get_a / coeffMtr hand out)
    void syncToHost() const
    {
        flush();
        chk(ipcgpu_linsys_get_values(ctx, const_cast<double*>(Base::a.data())));
    }

    void set_pattern(const std::vector<std::set<int>>& vNeighbor, const std::set<int>& fixedVert) override
    {
        Base::set_pattern(vNeighbor, fixedVert); // ia / ja 1-based, IJ2aI, a (LinSysSolver.hpp:46-150)
        // 0-based like CHOLMODSolver.cpp:76-77 (kept 0-based afterwards: load / write handle both, LinSysSolver.hpp:166-169)
        for (long i = 0; i < Base::ia.size(); ++i) Base::ia[i] -= 1;
        for (long i = 0; i < Base::ja.size(); ++i) Base::ja[i] -= 1;
        hipSetPattern(ctx, Base::numRows, Base::ia.data(), Base::ja.data());
        const size_t nnz = (size_t)Base::ja.size();
        delta_.assign(nnz, 0.0);
        setVal_.assign(nnz, 0.0);
        isSet_.assign(nnz, 0);
        dirty_ = anySet_ = false;
        Base::a.setZero();
    }
    void analyze_pattern(void) override { chk(ipcgpu_linsys_analyze_pattern(ctx)); }
    bool factorize(void) override
    {
        flush();
        const int rc = ipcgpu_linsys_factorize(ctx);
        chk(rc);
        return rc != IPCGPU_NOT_PD; // CHOLMODSolver.cpp:136
    }
    void solve(Eigen::VectorXd& rhs, Eigen::VectorXd& result) override
    {
        result.conservativeResize(rhs.size());
        chk(ipcgpu_linsys_solve(ctx, rhs.data(), result.data()));
    }
    void multiply(const Eigen::VectorXd& x, Eigen::VectorXd& Ax) override
    {
        flush();
        Ax.resize(x.size());
        chk(ipcgpu_linsys_multiply(ctx, x.data(), Ax.data()));
    }

    void setZero(void) override
    {
        std::fill(delta_.begin(), delta_.end(), 0.0);
        std::fill(isSet_.begin(), isSet_.end(), (unsigned char)0);
        dirty_ = anySet_ = false;
        chk(ipcgpu_linsys_set_zero(ctx));
    }
    void addCoeff(int rowI, int colI, double val) override
    {
        if (rowI <= colI) { // lower-triangle writes are ignored (LinSysSolver.hpp:404)
            delta_[slot(rowI, colI)] += val;
            dirty_ = true;
        }
    }
    void setCoeff(int rowI, int colI, double val) override
    {
        if (rowI <= colI) {
            const int k = slot(rowI, colI);
            isSet_[k] = 1;
            setVal_[k] = val;
            delta_[k] = 0.0;
            dirty_ = anySet_ = true;
        }
    }
    void setUnit_row(int rowI) override
    {
        for (const auto& colIter : Base::IJ2aI[rowI]) setCoeff(rowI, colIter.first, colIter.first == rowI ? 1.0 : 0.0);
    }
    void setUnit_col(int colI, const std::set<int>& rowVIs) override
    {
        for (const auto& rowVI : rowVIs)
            for (int dimI = 0; dimI < DIM; ++dimI) {
                const int rowI = rowVI * DIM + dimI;
                if (rowI <= colI && Base::IJ2aI[rowI].count(colI)) setCoeff(rowI, colI, rowI == colI ? 1.0 : 0.0);
            }
    }
    double coeffMtr(int rowI, int colI) const override
    {
        syncToHost();
        return Base::coeffMtr(rowI, colI);
    }
    void precondition_diag(const Eigen::VectorXd& input, Eigen::VectorXd& output) override
    {
        flush();
        output.resize(input.size());
        chk(ipcgpu_linsys_precondition_diag(ctx, input.data(), output.data()));
    }
    void getMaxDiag(double& maxDiag) override
    {
        syncToHost();
        Base::getMaxDiag(maxDiag);
    }
    Eigen::VectorXd& get_a(void) override
    {
        syncToHost();
        return Base::a;
    }
    const Eigen::VectorXd& get_a(void) const override
    {
        syncToHost();
        return Base::a;
    }
    // after the caller modified the array get_a() handed out (e.g. `linSys->get_a() += ...`): host mirror -> HBM
    void uploadFromHost()
    {
        flush();
        chk(ipcgpu_linsys_set_values(ctx, Base::a.data()));
    }
};

} // namespace IPC
