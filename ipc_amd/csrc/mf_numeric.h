// GPU multifrontal LL^T: numeric factorisation and triangular solves on the assembly tree produced by
// mf_symbolic.  Plays the role of cholmod_factorize / cholmod_solve (CHOLMODSolver.cpp:130-154).
#pragma once
#include "common.h"
#include "mf_symbolic.h"
#include <rocblas/rocblas.h>

namespace ipcgpu {

class MfNumeric {
public:
    MfNumeric() = default;
    ~MfNumeric();
    MfNumeric(const MfNumeric&) = delete;
    MfNumeric& operator=(const MfNumeric&) = delete;

    void setup(const MfSymbolic& sym, hipStream_t stream); // uploads maps, allocates fronts
    // a_dev: CSR values (device).  Returns false when a non-positive pivot was met.
    bool factorize(const double* a_dev);
    // rhs_dev / x_dev: device vectors in the user's ordering
    void solve(const double* rhs_dev, double* x_dev);
    bool ready() const { return ns_ > 0; }
    size_t front_bytes() const { return fronts_.n * sizeof(double); }

private:
    const MfSymbolic* sym_ = nullptr;
    hipStream_t stream_ = nullptr;
    rocblas_handle blas_ = nullptr;
    int ns_ = 0, nLevels_ = 0;
    DevBuf<double> fronts_, w_, yperm_;
    DevBuf<int> idx_, idxPtr_, firstNode_, childPtr_, child_, invPtr_, inv_, newOf_, levelFronts_, flag_, info_;
    DevBuf<long long> frontOff_, wOff_, aDst_;
    // extend-add work descriptors per level: (front, chunk)
    DevBuf<int2> eaDesc_;
    std::vector<int> eaLevelPtr_;
    // per level: fronts handled by the single-workgroup kernel and those sent to rocBLAS / rocSOLVER
    std::vector<std::vector<int>> smallFronts_, bigFronts_;
    DevBuf<int> smallList_;
    std::vector<int> smallLevelPtr_;
    PinnedBuf<int> hflag_;
    size_t ldsBytes_ = 0;
};

} // namespace ipcgpu
