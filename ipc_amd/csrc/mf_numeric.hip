// GPU multifrontal Cholesky, numeric phase (gfx950, fp64).
//
// Data layout in HBM: every front is a full N x N column-major square (ld = N) inside one buffer; the leading
// nc columns become the factor panel [L11; L21], the trailing (N-nc)^2 block is the update matrix that the
// parent gathers.  All fronts stay resident (sized for 288 GB of HBM3E: ~0.5 GB for a 45 K-node sheet), so
// there is no stack management and a child update is read in place.
// Scheduling: the assembly tree is processed level by level; inside a level fronts are independent.
//   extend-add   gather formulation (each parent entry sums its children through inverse index maps):
//                race-free and bit-reproducible, no atomics
//   factor       fronts with N <= bigN: one 256-thread workgroup per front, 32-column panels staged in LDS
//                (wave-level shuffle Cholesky of the 32x32 pivot block, per-row TRSM, 4x4 register tiles for
//                the Schur update); larger fronts: rocSOLVER dpotrf + rocBLAS dtrsm/dsyrk in place
//   solve        per-level forward / backward substitution, one workgroup per front, vectors in LDS
#include "mf_numeric.h"
#include <rocsolver/rocsolver.h>
#include <algorithm>
#include <cstdlib>

namespace ipcgpu {

namespace {

constexpr int NB = 32;
constexpr int WG = 256;
constexpr int EA_ITEMS = 8; // entries per thread in the extend-add kernel

struct TreeView {
    const long long* frontOff;
    const int* idxPtr;
    const int* firstNode;
    const int* childPtr;
    const int* child;
    const int* invPtr;
    const int* inv;
    const int* idx;
};

__global__ void k_scatter_a(int nnz, const double* __restrict__ a, const long long* __restrict__ dst, double* __restrict__ fronts)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) fronts[dst[k]] = a[k];
}

__global__ __launch_bounds__(WG) void k_extend_add(const int2* __restrict__ desc, TreeView tv, double* __restrict__ fronts)
{
    const int2 d = desc[blockIdx.x];
    const int s = d.x;
    const int N = 3 * (tv.idxPtr[s + 1] - tv.idxPtr[s]);
    const long long total = (long long)N * N;
    double* F = fronts + tv.frontOff[s];
    const int c0 = tv.childPtr[s], c1 = tv.childPtr[s + 1];
    long long e = (long long)d.y * (WG * EA_ITEMS) + threadIdx.x;
#pragma unroll 1
    for (int it = 0; it < EA_ITEMS; ++it, e += WG) {
        if (e >= total) break;
        const int J = (int)(e / N), I = (int)(e - (long long)J * N);
        if (I < J) continue;
        const int In = I / 3, Id = I - 3 * In, Jn = J / 3, Jd = J - 3 * Jn;
        double sum = 0.0;
        for (int ci = c0; ci < c1; ++ci) {
            const int c = tv.child[ci];
            const int* inv = tv.inv + tv.invPtr[c];
            const int ic = inv[In], jc = inv[Jn];
            if (ic >= 0 && jc >= 0) {
                const int Nc = 3 * (tv.idxPtr[c + 1] - tv.idxPtr[c]);
                const int ncc = 3 * (tv.firstNode[c + 1] - tv.firstNode[c]);
                sum += fronts[tv.frontOff[c] + (ncc + 3 * ic + Id) + (long long)Nc * (ncc + 3 * jc + Jd)];
            }
        }
        F[I + (long long)N * J] += sum;
    }
}

// One workgroup factors the leading nc columns of one front and forms its Schur complement in place.
__global__ __launch_bounds__(WG) void k_factor_front(const int* __restrict__ list, TreeView tv, double* __restrict__ fronts,
    int* __restrict__ flag)
{
    extern __shared__ double P[]; // NB panel columns, k-major: P[k * m + r]
    const int s = list[blockIdx.x];
    const int N = 3 * (tv.idxPtr[s + 1] - tv.idxPtr[s]);
    const int nc = 3 * (tv.firstNode[s + 1] - tv.firstNode[s]);
    double* F = fronts + tv.frontOff[s];
    const int tid = threadIdx.x;
    bool bad = false;

    for (int kb = 0; kb < nc; kb += NB) {
        const int w = min(NB, nc - kb);
        const int m = N - kb;
        // (a) stage the block column
        for (int e = tid; e < w * m; e += WG) {
            const int k = e / m, r = e - k * m;
            P[e] = (r >= k) ? F[(kb + r) + (long long)N * (kb + k)] : 0.0;
        }
        __syncthreads();
        // (b1) pivot block: lane r of wave 0 owns row r, columns live in registers, cross-lane reads by shuffle
        if (tid < 64) {
            double row[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) row[k] = (tid < w && k <= tid && k < w) ? P[k * m + tid] : 0.0;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (j < w) {
                    double djj = __shfl(row[j], j, 64);
                    if (!(djj > 0.0)) {
                        bad = true;
                        djj = 1.0;
                    }
                    const double dd = sqrt(djj);
                    const double invd = 1.0 / dd;
                    if (tid == j) row[j] = dd;
                    else if (tid > j) row[j] *= invd;
#pragma unroll
                    for (int jj = j + 1; jj < NB; ++jj) {
                        const double ljj = __shfl(row[j], jj, 64);
                        if (tid >= jj) row[jj] -= row[j] * ljj;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (tid < w && k <= tid && k < w) P[k * m + tid] = row[k];
        }
        __syncthreads();
        // (b2) rows below the pivot block: X L11^T = A21, one row per thread
        for (int r = w + tid; r < m; r += WG) {
            double x[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                if (k < w) {
                    double acc = P[k * m + r];
#pragma unroll
                    for (int q = 0; q < k; ++q) acc -= x[q] * P[q * m + k];
                    x[k] = acc / P[k * m + k];
                }
                else x[k] = 0.0;
            }
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (k < w) P[k * m + r] = x[k];
        }
        __syncthreads();
        // (c) factor panel back to HBM
        for (int e = tid; e < w * m; e += WG) {
            const int k = e / m, r = e - k * m;
            if (r >= k) F[(kb + r) + (long long)N * (kb + k)] = P[e];
        }
        // (d) Schur update of everything to the right: 4x4 register tiles, operands from LDS
        const int mt = m - w;
        const int ntile = (mt + 3) >> 2;
        const int ty = tid & 15, tx = tid >> 4;
        for (int tc = tx; tc < ntile; tc += 16) {
            for (int tr = ty; tr < ntile; tr += 16) {
                if (tr < tc) continue;
                const int i0 = w + 4 * tr, j0 = w + 4 * tc;
                double acc[4][4];
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) acc[ii][jj] = 0.0;
                for (int k = 0; k < w; ++k) {
                    const double* pk = P + k * m;
                    double av[4], bv[4];
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        av[ii] = pk[i0 + ii];
                        bv[ii] = pk[j0 + ii];
                    }
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) acc[ii][jj] += av[ii] * bv[jj];
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int j = j0 + jj;
                    if (j >= m) continue;
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const int i = i0 + ii;
                        if (i < m && i >= j) F[(kb + i) + (long long)N * (kb + j)] -= acc[ii][jj];
                    }
                }
            }
        }
        __syncthreads();
    }
    if (bad) atomicOr(flag, 1);
}

__global__ void k_permute_rhs(int nn, const int* __restrict__ newOf, const double* __restrict__ b, double* __restrict__ bp)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * nn) {
        int v = i / 3, d = i - 3 * v;
        bp[3 * newOf[v] + d] = b[i];
    }
}
__global__ void k_unpermute_x(int nn, const int* __restrict__ newOf, const double* __restrict__ xp, double* __restrict__ x)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * nn) {
        int v = i / 3, d = i - 3 * v;
        x[i] = xp[3 * newOf[v] + d];
    }
}

__global__ __launch_bounds__(WG) void k_fwd_level(const int* __restrict__ list, TreeView tv, const long long* __restrict__ wOff,
    const double* __restrict__ fronts, double* __restrict__ wbuf, double* __restrict__ yperm)
{
    extern __shared__ double w[];
    const int s = list[blockIdx.x];
    const int N = 3 * (tv.idxPtr[s + 1] - tv.idxPtr[s]);
    const int nc = 3 * (tv.firstNode[s + 1] - tv.firstNode[s]);
    const double* L = fronts + tv.frontOff[s];
    const int tid = threadIdx.x;
    const int col0 = 3 * tv.firstNode[s];
    const int c0 = tv.childPtr[s], c1 = tv.childPtr[s + 1];
    for (int I = tid; I < N; I += WG) {
        double val = (I < nc) ? yperm[col0 + I] : 0.0;
        const int In = I / 3, Id = I - 3 * In;
        for (int ci = c0; ci < c1; ++ci) {
            const int c = tv.child[ci];
            const int ic = tv.inv[tv.invPtr[c] + In];
            if (ic >= 0) {
                const int ncc = 3 * (tv.firstNode[c + 1] - tv.firstNode[c]);
                val += wbuf[wOff[c] + ncc + 3 * ic + Id];
            }
        }
        w[I] = val;
    }
    __syncthreads();
    for (int kb = 0; kb < nc; kb += NB) {
        const int wd = min(NB, nc - kb);
        if (tid < 64) {
            double y = (tid < wd) ? w[kb + tid] : 0.0;
            for (int j = 0; j < wd; ++j) {
                const double Ljj = L[(kb + j) + (long long)N * (kb + j)];
                const double yj = __shfl(y, j, 64) / Ljj;
                if (tid == j) y = yj;
                else if (tid > j && tid < wd) y -= L[(kb + tid) + (long long)N * (kb + j)] * yj;
            }
            if (tid < wd) w[kb + tid] = y;
        }
        __syncthreads();
        for (int i = kb + wd + tid; i < N; i += WG) {
            double acc = 0.0;
            for (int k = 0; k < wd; ++k) acc += L[i + (long long)N * (kb + k)] * w[kb + k];
            w[i] -= acc;
        }
        __syncthreads();
    }
    double* wo = wbuf + wOff[s];
    for (int I = tid; I < N; I += WG) {
        wo[I] = w[I]; // rows >= nc carry (children contributions - L21 y) up to the parent
        if (I < nc) yperm[col0 + I] = w[I];
    }
}

__global__ __launch_bounds__(WG) void k_bwd_level(const int* __restrict__ list, TreeView tv, const double* __restrict__ fronts,
    double* __restrict__ xperm)
{
    extern __shared__ double x[];
    const int s = list[blockIdx.x];
    const int N = 3 * (tv.idxPtr[s + 1] - tv.idxPtr[s]);
    const int nc = 3 * (tv.firstNode[s + 1] - tv.firstNode[s]);
    const double* L = fronts + tv.frontOff[s];
    const int* idx = tv.idx + tv.idxPtr[s];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int I = tid; I < N; I += WG) {
        const int In = I / 3;
        x[I] = xperm[3 * idx[In] + (I - 3 * In)];
    }
    __syncthreads();
    const int nblk = (nc + NB - 1) / NB;
    for (int b = nblk - 1; b >= 0; --b) {
        const int kb = b * NB;
        const int wd = min(NB, nc - kb);
        for (int j = wave; j < wd; j += WG / 64) {
            double acc = 0.0;
            const double* Lj = L + (long long)N * (kb + j);
            for (int i = kb + wd + lane; i < N; i += 64) acc += Lj[i] * x[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
            if (lane == 0) x[kb + j] -= acc;
        }
        __syncthreads();
        if (tid < 64) {
            double t = (tid < wd) ? x[kb + tid] : 0.0;
            for (int j = wd - 1; j >= 0; --j) {
                const double Ljj = L[(kb + j) + (long long)N * (kb + j)];
                const double xj = __shfl(t, j, 64) / Ljj;
                if (tid == j) t = xj;
                else if (tid < j) t -= L[(kb + j) + (long long)N * (kb + tid)] * xj;
            }
            if (tid < wd) x[kb + tid] = t;
        }
        __syncthreads();
    }
    const int col0 = 3 * tv.firstNode[s];
    for (int I = tid; I < nc; I += WG) xperm[col0 + I] = x[I];
}

} // namespace

MfNumeric::~MfNumeric()
{
    if (blas_) rocblas_destroy_handle(blas_);
}

void MfNumeric::setup(const MfSymbolic& sym, hipStream_t stream)
{
    sym_ = &sym;
    stream_ = stream;
    ns_ = sym.ns;
    nLevels_ = (int)sym.levelPtr.size() - 1;
    if (!blas_) {
        if (rocblas_create_handle(&blas_) != rocblas_status_success) throw HipError("rocblas_create_handle failed");
    }
    rocblas_set_stream(blas_, stream_);
    fronts_.alloc((size_t)sym.frontOff[ns_]);
    w_.alloc((size_t)sym.wOff[ns_]);
    yperm_.alloc((size_t)sym.n);
    idx_.upload(sym.idx, stream);
    idxPtr_.upload(sym.idxPtr, stream);
    firstNode_.upload(sym.firstNode, stream);
    childPtr_.upload(sym.childPtr, stream);
    child_.upload(sym.child.empty() ? std::vector<int>{ 0 } : sym.child, stream);
    invPtr_.upload(sym.invPtr, stream);
    inv_.upload(sym.inv.empty() ? std::vector<int>{ 0 } : sym.inv, stream);
    newOf_.upload(sym.newOf, stream);
    levelFronts_.upload(sym.levelFronts, stream);
    {
        std::vector<long long> t(sym.frontOff.begin(), sym.frontOff.end());
        frontOff_.upload(t, stream);
        std::vector<long long> u(sym.wOff.begin(), sym.wOff.end());
        wOff_.upload(u, stream);
        std::vector<long long> d(sym.aDst.begin(), sym.aDst.end());
        aDst_.upload(d, stream);
    }
    flag_.alloc(1);
    hflag_.alloc(4);
    // split fronts of every level into the single-workgroup class and the rocBLAS class
    int bigN = 448;
    if (const char* e = std::getenv("IPCGPU_MF_BIGN")) bigN = std::max(33, std::min(448, std::atoi(e)));
    smallFronts_.assign(nLevels_, {});
    bigFronts_.assign(nLevels_, {});
    std::vector<int> smallList;
    smallLevelPtr_.assign(nLevels_ + 1, 0);
    int maxSmallN = 0, nBig = 0;
    for (int l = 0; l < nLevels_; ++l) {
        for (int i = sym.levelPtr[l]; i < sym.levelPtr[l + 1]; ++i) {
            const int s = sym.levelFronts[i];
            if (sym.N(s) <= bigN) {
                smallFronts_[l].push_back(s);
                maxSmallN = std::max(maxSmallN, sym.N(s));
            }
            else {
                bigFronts_[l].push_back(s);
                ++nBig;
            }
        }
        // heaviest first so the tail of the level is made of short jobs
        std::sort(smallFronts_[l].begin(), smallFronts_[l].end(), [&](int a, int b) { return sym.N(a) > sym.N(b); });
        smallList.insert(smallList.end(), smallFronts_[l].begin(), smallFronts_[l].end());
        smallLevelPtr_[l + 1] = (int)smallList.size();
    }
    if (smallList.empty()) smallList.push_back(0);
    smallList_.upload(smallList, stream);
    info_.alloc(std::max(1, nBig));
    ldsBytes_ = (size_t)(NB * maxSmallN + 8) * sizeof(double);
    if (ldsBytes_ > 64 * 1024)
        HIP_CHECK(hipFuncSetAttribute((const void*)k_factor_front, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes_));
    const size_t solveLds = (size_t)sym.maxN * sizeof(double);
    if (solveLds > 64 * 1024) {
        if (solveLds > 160 * 1024) throw StateError("front too large for the in-LDS solve kernels");
        HIP_CHECK(hipFuncSetAttribute((const void*)k_fwd_level, hipFuncAttributeMaxDynamicSharedMemorySize, (int)solveLds));
        HIP_CHECK(hipFuncSetAttribute((const void*)k_bwd_level, hipFuncAttributeMaxDynamicSharedMemorySize, (int)solveLds));
    }
    // extend-add descriptors (levels >= 1 only; leaves have no children)
    std::vector<int2> desc;
    eaLevelPtr_.assign(nLevels_ + 1, 0);
    for (int l = 0; l < nLevels_; ++l) {
        for (int i = sym.levelPtr[l]; i < sym.levelPtr[l + 1]; ++i) {
            const int s = sym.levelFronts[i];
            if (sym.childPtr[s + 1] == sym.childPtr[s]) continue;
            const long long total = (long long)sym.N(s) * sym.N(s);
            const int chunks = (int)((total + WG * EA_ITEMS - 1) / (WG * EA_ITEMS));
            for (int c = 0; c < chunks; ++c) desc.push_back(make_int2(s, c));
        }
        eaLevelPtr_[l + 1] = (int)desc.size();
    }
    if (desc.empty()) desc.push_back(make_int2(0, 0));
    eaDesc_.upload(desc.data(), desc.size(), stream);
    HIP_CHECK(hipStreamSynchronize(stream));
}

bool MfNumeric::factorize(const double* a_dev)
{
    if (!sym_) throw StateError("factorize before analyze_pattern");
    const MfSymbolic& sym = *sym_;
    TreeView tv{ frontOff_.p, idxPtr_.p, firstNode_.p, childPtr_.p, child_.p, invPtr_.p, inv_.p, idx_.p };
    fronts_.zero(stream_);
    flag_.zero(stream_);
    info_.zero(stream_);
    const int nnz = (int)sym.aDst.size();
    hipLaunchKernelGGL(k_scatter_a, dim3((nnz + 255) / 256), dim3(256), 0, stream_, nnz, a_dev, aDst_.p, fronts_.p);
    int bigCount = 0;
    for (int l = 0; l < nLevels_; ++l) {
        const int nea = eaLevelPtr_[l + 1] - eaLevelPtr_[l];
        if (nea > 0) hipLaunchKernelGGL(k_extend_add, dim3(nea), dim3(WG), 0, stream_, eaDesc_.p + eaLevelPtr_[l], tv, fronts_.p);
        const int nsmall = smallLevelPtr_[l + 1] - smallLevelPtr_[l];
        if (nsmall > 0)
            hipLaunchKernelGGL(k_factor_front, dim3(nsmall), dim3(WG), ldsBytes_, stream_, smallList_.p + smallLevelPtr_[l], tv,
                fronts_.p, flag_.p);
        for (int s : bigFronts_[l]) {
            const int N = sym.N(s), nc = sym.nc(s), nb = N - nc;
            double* F = fronts_.p + sym.frontOff[s];
            const double one = 1.0, mone = -1.0;
            if (rocsolver_dpotrf(blas_, rocblas_fill_lower, nc, F, N, info_.p + bigCount) != rocblas_status_success)
                throw HipError("rocsolver_dpotrf failed");
            ++bigCount;
            if (nb > 0) {
                if (rocblas_dtrsm(blas_, rocblas_side_right, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, nb,
                        nc, &one, F, N, F + nc, N)
                    != rocblas_status_success)
                    throw HipError("rocblas_dtrsm failed");
                if (rocblas_dsyrk(blas_, rocblas_fill_lower, rocblas_operation_none, nb, nc, &mone, F + nc, N, &one,
                        F + nc + (size_t)N * nc, N)
                    != rocblas_status_success)
                    throw HipError("rocblas_dsyrk failed");
            }
        }
    }
    // not-PD detection: flag from the workgroup kernel, info[] from rocSOLVER
    HIP_CHECK(hipMemcpyAsync(hflag_.p, flag_.p, sizeof(int), hipMemcpyDeviceToHost, stream_));
    std::vector<int> hinfo(std::max(1, bigCount), 0);
    if (bigCount) HIP_CHECK(hipMemcpyAsync(hinfo.data(), info_.p, sizeof(int) * bigCount, hipMemcpyDeviceToHost, stream_));
    HIP_CHECK(hipStreamSynchronize(stream_));
    bool ok = (hflag_.p[0] == 0);
    for (int i = 0; i < bigCount; ++i) ok = ok && (hinfo[i] == 0);
    return ok;
}

void MfNumeric::solve(const double* rhs_dev, double* x_dev)
{
    if (!sym_) throw StateError("solve before analyze_pattern");
    const MfSymbolic& sym = *sym_;
    TreeView tv{ frontOff_.p, idxPtr_.p, firstNode_.p, childPtr_.p, child_.p, invPtr_.p, inv_.p, idx_.p };
    const int n3 = sym.n;
    hipLaunchKernelGGL(k_permute_rhs, dim3((n3 + 255) / 256), dim3(256), 0, stream_, sym.nn, newOf_.p, rhs_dev, yperm_.p);
    const size_t lds = (size_t)sym.maxN * sizeof(double);
    for (int l = 0; l < nLevels_; ++l) {
        const int cnt = sym.levelPtr[l + 1] - sym.levelPtr[l];
        if (cnt > 0)
            hipLaunchKernelGGL(k_fwd_level, dim3(cnt), dim3(WG), lds, stream_, levelFronts_.p + sym.levelPtr[l], tv, wOff_.p, fronts_.p,
                w_.p, yperm_.p);
    }
    for (int l = nLevels_ - 1; l >= 0; --l) {
        const int cnt = sym.levelPtr[l + 1] - sym.levelPtr[l];
        if (cnt > 0)
            hipLaunchKernelGGL(k_bwd_level, dim3(cnt), dim3(WG), lds, stream_, levelFronts_.p + sym.levelPtr[l], tv, fronts_.p, yperm_.p);
    }
    hipLaunchKernelGGL(k_unpermute_x, dim3((n3 + 255) / 256), dim3(256), 0, stream_, sym.nn, newOf_.p, yperm_.p, x_dev);
}

} // namespace ipcgpu
