// HipLinSysSolver: symmetric-upper CSR with 3x3 node blocks (LinSysSolver.hpp:46-150) resident in HBM,
// factorised by the GPU multifrontal Cholesky (default) or by rocSOLVER's csrrf re-factorisation.
#include "hip_ipc.h"
#include <rocsolver/rocsolver.h>
#include <algorithm>
#include <thread>
#include <cstdlib>

namespace ipcgpu {

// ---- rocSOLVER csrrf back end -------------------------------------------------------------------------
// rocsolver_dcsrrf_refactchol re-factorises on a *given* ordering Q and pattern of L (T); both come from
// mf_symbolic (rocSOLVER ships no symbolic phase).  It reads the LOWER CSR triangle of the matrix in the
// original ordering (it applies Q itself), while the reference stores the UPPER CSR triangle, i.e. the
// lower CSC triangle.  One static transpose permutation per pattern converts between the two.
struct RocsolverCsrrf {
    rocblas_handle handle = nullptr;
    rocsolver_rfinfo info = nullptr;
    int n = 0, nnzA = 0, nnzT = 0;
    DevBuf<int> ptrA, indA, ptrT, indT, pivQ, trans, flag; // trans[k] = index into the upper-CSR values for lower-CSR slot k
    DevBuf<double> valA, valT, B;
    bool analyzed = false;
    ~RocsolverCsrrf()
    {
        if (info) rocsolver_destroy_rfinfo(info);
        if (handle) rocblas_destroy_handle(handle);
    }
};

namespace {
__global__ void k_gather(int n, const int* __restrict__ map, const double* __restrict__ src, double* __restrict__ dst)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[map[i]];
}
__global__ void k_set_diag_one(int n, const int* __restrict__ ptrT, const int* __restrict__ indT, double* __restrict__ valT)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) {
        for (int k = ptrT[r]; k < ptrT[r + 1]; ++k) valT[k] = (indT[k] == r) ? 1.0 : 0.0;
    }
}
// rocSPARSE's csric0 (behind csrrf_refactchol) does not surface a non-positive pivot: it returns success with a
// finite factor.  A genuine Cholesky factor satisfies (L L^T)_rr = A_rr row by row; a pivot whose sign was lost
// breaks that identity, so the check is one pass over T.
__global__ void k_check_diag(int n, const int* __restrict__ ptrT, const double* __restrict__ valT, const int* __restrict__ pivQ,
    const int* __restrict__ ia, const double* __restrict__ a, int* __restrict__ flag)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) {
        double s2 = 0.0;
        for (int k = ptrT[r]; k < ptrT[r + 1]; ++k) s2 += valT[k] * valT[k];
        const double arr = a[ia[pivQ[r]]]; // the diagonal leads every upper-CSR row
        const double d = valT[ptrT[r + 1] - 1];
        if (!(d > 0.0) || !(fabs(s2 - arr) <= 1e-8 * fabs(arr) + 1e-300)) atomicOr(flag, 1);
    }
}
} // namespace

HipLinSysSolver::HipLinSysSolver(hipStream_t s) : stream(s) {}
HipLinSysSolver::~HipLinSysSolver() = default;

void HipLinSysSolver::set_pattern(const HipMesh& mesh, int nExtra, const int* extraPairs)
{
    const int nV = mesh.nV;
    // neighbours > v, ascending: mesh vNeighbor merged with the extra (contact) pairs.  Flat CSR-like storage (counting pass,
    // fill pass, per-node sort + unique) and a few threads over node ranges: this runs on every change of the contact pattern.
    for (int i = 0; i < nExtra; ++i) {
        const int a = extraPairs[2 * i], b = extraPairs[2 * i + 1];
        if (a < 0 || b < 0 || a >= nV || b >= nV) throw ArgError("set_pattern: extra pair out of range");
    }
    std::vector<int> upPtr(nV + 1, 0);
    for (int v = 0; v < nV; ++v)
        for (int k = mesh.nbPtr[v]; k < mesh.nbPtr[v + 1]; ++k)
            if (mesh.nb[k] > v) upPtr[v + 1]++;
    for (int i = 0; i < nExtra; ++i) {
        const int a = extraPairs[2 * i], b = extraPairs[2 * i + 1];
        if (a != b) upPtr[std::min(a, b) + 1]++;
    }
    for (int v = 0; v < nV; ++v) upPtr[v + 1] += upPtr[v];
    std::vector<int> upRaw(upPtr[nV]), upCnt(nV, 0);
    {
        std::vector<int> pos(upPtr.begin(), upPtr.end() - 1);
        for (int v = 0; v < nV; ++v)
            for (int k = mesh.nbPtr[v]; k < mesh.nbPtr[v + 1]; ++k)
                if (mesh.nb[k] > v) upRaw[pos[v]++] = mesh.nb[k];
        for (int i = 0; i < nExtra; ++i) {
            const int a = extraPairs[2 * i], b = extraPairs[2 * i + 1];
            if (a != b) upRaw[pos[std::min(a, b)]++] = std::max(a, b);
        }
    }
    const int nThreads = std::max(1, std::min(8, std::min((int)std::thread::hardware_concurrency(), nV / 4096 + 1)));
    auto parallel = [&](auto&& body) {
        if (nThreads == 1) {
            body(0, nV);
            return;
        }
        std::vector<std::thread> pool;
        for (int t = 0; t < nThreads; ++t)
            pool.emplace_back([&, t] { body((int)((long long)nV * t / nThreads), (int)((long long)nV * (t + 1) / nThreads)); });
        for (auto& th : pool) th.join();
    };
    parallel([&](int v0, int v1) {
        for (int v = v0; v < v1; ++v) {
            int* b = upRaw.data() + upPtr[v];
            int* e = upRaw.data() + upPtr[v + 1];
            std::sort(b, e);
            upCnt[v] = (int)(std::unique(b, e) - b);
        }
    });
    numRows = 3 * nV;
    ia.assign(numRows + 1, 0);
    rowBase.assign(nV, 0);
    rowLen.assign(nV, 0);
    for (int v = 0; v < nV; ++v) {
        const int nnz = 3 + 3 * upCnt[v]; // LinSysSolver.hpp:63-111: row 3v has nnz, 3v+1 nnz-1, 3v+2 nnz-2 entries
        ia[3 * v + 1] = ia[3 * v] + nnz;
        ia[3 * v + 2] = ia[3 * v + 1] + nnz - 1;
        ia[3 * v + 3] = ia[3 * v + 2] + nnz - 2;
        rowBase[v] = ia[3 * v];
        rowLen[v] = nnz;
    }
    ja.resize(ia[numRows]);
    parallel([&](int v0, int v1) {
        for (int v = v0; v < v1; ++v) {
            const int* u = upRaw.data() + upPtr[v];
            for (int r = 0; r < 3; ++r) {
                int p = ia[3 * v + r];
                for (int c = r; c < 3; ++c) ja[p++] = 3 * v + c;
                for (int q = 0; q < upCnt[v]; ++q)
                    for (int c = 0; c < 3; ++c) ja[p++] = 3 * u[q] + c;
            }
        }
    });
    d_ia.uploadGrow(ia, stream);
    d_ja.uploadGrow(ja, stream);
    d_a.ensure(ja.size());
    d_a.zeroN(ja.size(), stream);
    d_rowBase.uploadGrow(rowBase, stream);
    d_rowLen.uploadGrow(rowLen, stream);
    // tet edge -> first slot of its 3x3 block in row 3*min: ia[3 vmin] + 3 + 3 * rank(vmax among up[vmin])
    std::vector<int> edgeP0(6 * (size_t)mesh.nT);
    static const int ea[6] = { 0, 0, 0, 1, 1, 2 }, eb[6] = { 1, 2, 3, 2, 3, 3 };
    {
        const int nT = mesh.nT;
        auto edges = [&](int t0, int t1) {
            for (int t = t0; t < t1; ++t)
                for (int e = 0; e < 6; ++e) {
                    const int va = mesh.F[t + (size_t)nT * ea[e]], vb = mesh.F[t + (size_t)nT * eb[e]];
                    const int lo = std::min(va, vb), hi = std::max(va, vb);
                    const int* ub = upRaw.data() + upPtr[lo];
                    const int rank = int(std::lower_bound(ub, ub + upCnt[lo], hi) - ub);
                    edgeP0[(size_t)e * nT + t] = rowBase[lo] + 3 + 3 * rank;
                }
        };
        if (nThreads == 1 || nT < 8192) edges(0, nT);
        else {
            std::vector<std::thread> pool;
            for (int t = 0; t < nThreads; ++t)
                pool.emplace_back([&, t] { edges((int)((long long)nT * t / nThreads), (int)((long long)nT * (t + 1) / nThreads)); });
            for (auto& th : pool) th.join();
        }
    }
    d_edgeP0.uploadGrow(edgeP0, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
    analyzed_ = false;
    ++patternVersion;
}

void HipLinSysSolver::set_pattern_csr(int nRows, const int* ia_, const int* ja_)
{
    if (nRows <= 0 || nRows % 3) throw ArgError("set_pattern_csr: rows must be a positive multiple of 3");
    numRows = nRows;
    ia.assign(ia_, ia_ + nRows + 1);
    ja.assign(ja_, ja_ + ia[nRows]);
    for (int r = 0; r < nRows; ++r) {
        if (ia[r + 1] <= ia[r] || ja[ia[r]] != r) throw ArgError("set_pattern_csr: every row must start with its diagonal (upper CSR)");
        for (int k = ia[r] + 1; k < ia[r + 1]; ++k)
            if (ja[k] <= ja[k - 1] || ja[k] >= nRows) throw ArgError("set_pattern_csr: columns must be ascending and in range");
    }
    d_ia.upload(ia, stream);
    d_ja.upload(ja, stream);
    d_a.alloc(ja.size());
    d_a.zero(stream);
    rowBase.clear();
    rowLen.clear();
    HIP_CHECK(hipStreamSynchronize(stream));
    analyzed_ = false;
    ++patternVersion;
}

void HipLinSysSolver::setZero() { d_a.zeroN(ja.size(), stream); }

int HipLinSysSolver::findEntry(int row, int col) const
{
    if (row < 0 || row >= numRows) return -1;
    const int* b = ja.data() + ia[row];
    const int* e = ja.data() + ia[row + 1];
    const int* it = std::lower_bound(b, e, col);
    if (it == e || *it != col) return -1;
    return int(it - ja.data());
}

void HipLinSysSolver::analyze_pattern(const HipMesh* mesh)
{
    if (!numRows) throw StateError("analyze_pattern before set_pattern");
    std::vector<double> coords;
    const double* cptr = nullptr;
    if (mesh && 3 * mesh->nV == numRows) {
        coords.resize(3 * (size_t)mesh->nV);
        for (int v = 0; v < mesh->nV; ++v)
            for (int c = 0; c < 3; ++c) coords[3 * (size_t)v + c] = mesh->V_rest[v + (size_t)mesh->nV * c];
        cptr = coords.data();
    }
    // leaf domains of at most 12 nodes (36 columns): one level less at the bottom of the tree than with 8 -- one launch less in the factorisation and in each sweep --
    // and still a single-workgroup front in 64 KB of LDS.  Measured (profiles/r04_nd_leaf_size_ab.txt): mat150 5: 382, 6: 384, 8: 393, 10: 396, 12: 406, 14: 406, 16: 394,
    // 20: 398 it/s; mat433 42.9 -> 43.7; contact bench 10.98 -> 10.78 ms per iteration.
    const int leaf = 12;
    // (the entries' destinations in the fronts are computed on the device by MfNumeric::setup; the rocSOLVER back end does not use them)
    mf_analyze(numRows, ia.data(), ja.data(), cptr, leaf, sym_, /*withEntryDestinations=*/false);
    ++analysisVersion;
    if (solverType == 0) {
        num_.setup(sym_, stream, d_ia.p, d_ja.p, (long long)ja.size());
    }
    else {
        rs_.reset(new RocsolverCsrrf);
        RocsolverCsrrf& R = *rs_;
        R.n = numRows;
        if (rocblas_create_handle(&R.handle) != rocblas_status_success) throw HipError("rocblas_create_handle");
        rocblas_set_stream(R.handle, stream);
        if (rocsolver_create_rfinfo(&R.info, R.handle) != rocblas_status_success) throw HipError("rocsolver_create_rfinfo");
        if (rocsolver_set_rfinfo_mode(R.info, rocsolver_rfinfo_mode_cholesky) != rocblas_status_success)
            throw HipError("rocsolver_set_rfinfo_mode");
        // lower CSR of A (original ordering) = transpose of the upper CSR
        const int nnz = (int)ja.size();
        std::vector<int> ptr(numRows + 1, 0), ind(nnz), tr(nnz);
        for (int k = 0; k < nnz; ++k) ptr[ja[k] + 1]++;
        for (int r = 0; r < numRows; ++r) ptr[r + 1] += ptr[r];
        std::vector<int> pos(ptr.begin(), ptr.end() - 1);
        for (int r = 0; r < numRows; ++r)
            for (int k = ia[r]; k < ia[r + 1]; ++k) {
                const int c = ja[k];
                ind[pos[c]] = r;
                tr[pos[c]] = k;
                pos[c]++;
            }
        std::vector<int> pT, iT, q;
        mf_L_pattern_csr(sym_, pT, iT, q);
        R.nnzA = nnz;
        R.nnzT = (int)iT.size();
        R.ptrA.upload(ptr, stream);
        R.indA.upload(ind, stream);
        R.trans.upload(tr, stream);
        R.valA.alloc(nnz);
        R.ptrT.upload(pT, stream);
        R.indT.upload(iT, stream);
        R.pivQ.upload(q, stream);
        R.valT.alloc(iT.size());
        R.B.alloc(numRows);
        R.B.zero(stream);
        R.flag.alloc(1);
        // the analysis wants a numerically valid pair (M, T): use M = I-pattern values, T = identity factor
        hipLaunchKernelGGL(k_set_diag_one, dim3((numRows + 255) / 256), dim3(256), 0, stream, numRows, R.ptrT.p, R.indT.p, R.valT.p);
        hipLaunchKernelGGL(k_set_diag_one, dim3((numRows + 255) / 256), dim3(256), 0, stream, numRows, R.ptrA.p, R.indA.p, R.valA.p);
        const rocblas_status st = rocsolver_dcsrrf_analysis(R.handle, R.n, 1, R.nnzA, R.ptrA.p, R.indA.p, R.valA.p, R.nnzT, R.ptrT.p, R.indT.p, R.valT.p,
            nullptr, R.pivQ.p, R.B.p, R.n, R.info);
        if (st != rocblas_status_success) throw HipError("rocsolver_dcsrrf_analysis failed (rocblas_status " + std::to_string((int)st) + ")");
        HIP_CHECK(hipStreamSynchronize(stream));
        R.analyzed = true;
    }
    analyzed_ = true;
}

bool HipLinSysSolver::factorize()
{
    if (!analyzed_) throw StateError("factorize before analyze_pattern");
    if (solverType == 0) return num_.factorize(d_a.p);
    RocsolverCsrrf& R = *rs_;
    hipLaunchKernelGGL(k_gather, dim3((R.nnzA + 255) / 256), dim3(256), 0, stream, R.nnzA, R.trans.p, d_a.p, R.valA.p);
    rocblas_status st = rocsolver_dcsrrf_refactchol(R.handle, R.n, R.nnzA, R.ptrA.p, R.indA.p, R.valA.p, R.nnzT, R.ptrT.p, R.indT.p,
        R.valT.p, R.pivQ.p, R.info);
    if (st != rocblas_status_success) {
        HIP_CHECK(hipStreamSynchronize(stream));
        return false;
    }
    R.flag.zero(stream);
    hipLaunchKernelGGL(k_check_diag, dim3((R.n + 255) / 256), dim3(256), 0, stream, R.n, R.ptrT.p, R.valT.p, R.pivQ.p, d_ia.p, d_a.p,
        R.flag.p);
    int f = 0;
    HIP_CHECK(hipMemcpyAsync(&f, R.flag.p, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    return f == 0;
}

// factorize() followed by solve(), with the forward sweep running beside the factorisation (MfNumeric::factorizeSolve)
bool HipLinSysSolver::factorizeSolve(const double* rhs_dev, double* x_dev, bool wait)
{
    if (!analyzed_) throw StateError("factorize before analyze_pattern");
    lastSyncOk_ = true;
    if (solverType == 0) return lastSyncOk_ = num_.factorizeSolve(d_a.p, rhs_dev, x_dev, wait);
    const bool ok = factorize();
    if (ok) solve(rhs_dev, x_dev);
    return lastSyncOk_ = ok;
}
bool HipLinSysSolver::lastPivotsOk() const { return lastSyncOk_ && (solverType != 0 || num_.lastPivotsOk()); }

void HipLinSysSolver::solve(const double* rhs_dev, double* x_dev)
{
    if (!analyzed_) throw StateError("solve before analyze_pattern");
    if (solverType == 0) {
        num_.solve(rhs_dev, x_dev);
        return;
    }
    RocsolverCsrrf& R = *rs_;
    HIP_CHECK(hipMemcpyAsync(x_dev, rhs_dev, sizeof(double) * R.n, hipMemcpyDeviceToDevice, stream));
    if (rocsolver_dcsrrf_solve(R.handle, R.n, 1, R.nnzT, R.ptrT.p, R.indT.p, R.valT.p, nullptr, R.pivQ.p, x_dev, R.n, R.info)
        != rocblas_status_success)
        throw HipError("rocsolver_dcsrrf_solve failed");
}

void HipLinSysSolver::multiply(const double* x_dev, double* y_dev)
{
    launch_csr_symv(numRows, d_ia.p, d_ja.p, d_a.p, x_dev, y_dev, stream);
}
void HipLinSysSolver::precondition_diag(const double* in_dev, double* out_dev)
{
    launch_precond_diag(numRows, d_ia.p, d_a.p, in_dev, out_dev, stream);
}

} // namespace ipcgpu
