// GPU multifrontal Cholesky, numeric phase (gfx950, fp64).
//
// Data layout in HBM: every front is a full N x N column-major square (ld = N) inside one buffer; the leading
// nc columns become the factor panel [L11; L21], the trailing (N-nc)^2 block is the update matrix that the
// parent gathers.  All fronts stay resident (sized for 288 GB of HBM3E: ~0.5 GB for a 45 K-node sheet), so
// there is no stack management and a child update is read in place.
//
// Scheduling: the assembly tree is processed level by level; inside a level fronts are independent.
//   extend-add   gather formulation (each parent entry sums its children through inverse index maps):
//                race-free and bit-reproducible, no atomics
//   small fronts one 256-thread workgroup per front: 32-column panels staged in LDS, wave-level shuffle
//                Cholesky of the 32x32 pivot block, per-row TRSM, 4x4 register tiles for the Schur update
//   big fronts   level-batched 32-column steps, two launches per step for ALL big fronts of the level:
//                  k_big_trsm  every workgroup re-factors the 32x32 pivot block in registers (cheaper than a
//                              launch boundary) and solves its 256 rows of the panel
//                  k_big_syrk  64x64 tiles of the trailing matrix, panels staged in LDS, 4x4 register tiles;
//                              the first tile of a front also publishes the factored pivot block
//   solve        per-level forward / backward substitution: one workgroup per small front (vectors in LDS);
//                big fronts again in level-batched 32-column steps with all workgroups sharing the row updates
// No vendor BLAS is involved: rocSOLVER's potrf / rocBLAS' trsm+syrk cost ~150 tiny launches per front.
#include "mf_numeric.h"
#include <algorithm>
#include <cstdlib>

namespace ipcgpu {

namespace {

constexpr int NB = 32;
constexpr int WG = 256;
constexpr int EA_ITEMS = 8; // entries per thread in the extend-add kernel
constexpr int TS = 64; // trailing-update tile

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

__device__ __forceinline__ int frontN(const TreeView& tv, int s) { return 3 * (tv.idxPtr[s + 1] - tv.idxPtr[s]); }
__device__ __forceinline__ int frontNc(const TreeView& tv, int s) { return 3 * (tv.firstNode[s + 1] - tv.firstNode[s]); }

__global__ void k_scatter_a(int nnz, const double* __restrict__ a, const long long* __restrict__ dst, double* __restrict__ fronts)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) fronts[dst[k]] = a[k];
}

__global__ __launch_bounds__(WG) void k_extend_add(const int2* __restrict__ desc, TreeView tv, double* __restrict__ fronts)
{
    const int2 d = desc[blockIdx.x];
    const int s = d.x;
    const int N = frontN(tv, s);
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
                const int Nc = frontN(tv, c);
                const int ncc = frontNc(tv, c);
                sum += fronts[tv.frontOff[c] + (ncc + 3 * ic + Id) + (long long)Nc * (ncc + 3 * jc + Jd)];
            }
        }
        F[I + (long long)N * J] += sum;
    }
}

// Cholesky of a (<=) 32x32 pivot block by one wave: lane r owns row r in registers, cross-lane reads by shuffle.
// blk is k-major in LDS: blk[k * ld + r] = A(r, k).  Columns / rows >= w are ignored.  Returns true on a bad pivot.
__device__ __forceinline__ bool wave_potrf32(double* blk, int ld, int w, int lane)
{
    bool bad = false;
    double row[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) row[k] = (lane < w && k <= lane && k < w) ? blk[k * ld + lane] : 0.0;
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
            if (lane == j) row[j] = dd;
            else if (lane > j) row[j] *= invd;
#pragma unroll
            for (int jj = j + 1; jj < NB; ++jj) {
                const double ljj = __shfl(row[j], jj, 64);
                if (lane >= jj) row[jj] -= row[j] * ljj;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NB; ++k)
        if (lane < w && k <= lane && k < w) blk[k * ld + lane] = row[k];
    return bad;
}

// One workgroup factors the leading nc columns of one small front and forms its Schur complement in place.
__global__ __launch_bounds__(WG) void k_factor_front(const int* __restrict__ list, TreeView tv, double* __restrict__ fronts,
    int* __restrict__ flag)
{
    extern __shared__ double P[]; // NB panel columns, k-major: P[k * m + r]
    const int s = list[blockIdx.x];
    const int N = frontN(tv, s);
    const int nc = frontNc(tv, s);
    double* F = fronts + tv.frontOff[s];
    const int tid = threadIdx.x;
    bool bad = false;

    for (int kb = 0; kb < nc; kb += NB) {
        const int w = min(NB, nc - kb);
        const int m = N - kb;
        for (int e = tid; e < w * m; e += WG) {
            const int k = e / m, r = e - k * m;
            P[e] = (r >= k) ? F[(kb + r) + (long long)N * (kb + k)] : 0.0;
        }
        __syncthreads();
        if (tid < 64) bad |= wave_potrf32(P, m, w, tid);
        __syncthreads();
        // rows below the pivot block: X L11^T = A21, one row per thread
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
        for (int e = tid; e < w * m; e += WG) {
            const int k = e / m, r = e - k * m;
            if (r >= k) F[(kb + r) + (long long)N * (kb + k)] = P[e];
        }
        // Schur update of everything to the right: 4x4 register tiles, operands from LDS
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

// ---- big fronts: level-batched 32-column steps ---------------------------------------------------------
// desc = (front, kb, first row offset behind the pivot block, unused)
__global__ __launch_bounds__(WG) void k_big_trsm(const int4* __restrict__ desc, TreeView tv, double* __restrict__ fronts,
    int* __restrict__ flag)
{
    __shared__ double L11[NB * NB];
    const int4 d = desc[blockIdx.x];
    const int s = d.x, kb = d.y;
    const int N = frontN(tv, s), nc = frontNc(tv, s);
    const int w = min(NB, nc - kb);
    double* F = fronts + tv.frontOff[s];
    const int tid = threadIdx.x;
    for (int e = tid; e < NB * NB; e += WG) {
        const int k = e / NB, r = e - k * NB;
        L11[e] = (k < w && r < w && r >= k) ? F[(kb + r) + (long long)N * (kb + k)] : 0.0;
    }
    __syncthreads();
    if (tid < 64) {
        if (wave_potrf32(L11, NB, w, tid)) atomicOr(flag, 1);
    }
    __syncthreads();
    const int r = kb + w + d.z + tid;
    if (r < N) {
        double x[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            if (k < w) {
                double acc = F[r + (long long)N * (kb + k)];
#pragma unroll
                for (int q = 0; q < k; ++q) acc -= x[q] * L11[q * NB + k];
                x[k] = acc / L11[k * NB + k];
            }
            else x[k] = 0.0;
        }
#pragma unroll
        for (int k = 0; k < NB; ++k)
            if (k < w) F[r + (long long)N * (kb + k)] = x[k];
    }
}

// desc = (front, kb, ti, tj); ti < 0: no tile, publish the pivot block only; desc.w bit 30 of ti... kept simple:
// the workgroup with (ti == tj == 0) or (ti < 0) also factors and writes the pivot block L11 to HBM.
__global__ __launch_bounds__(WG) void k_big_syrk(const int4* __restrict__ desc, TreeView tv, double* __restrict__ fronts)
{
    __shared__ double As[NB][TS];
    __shared__ double Bs[NB][TS];
    const int4 d = desc[blockIdx.x];
    const int s = d.x, kb = d.y, ti = d.z, tj = d.w;
    const int N = frontN(tv, s), nc = frontNc(tv, s);
    const int w = min(NB, nc - kb);
    double* F = fronts + tv.frontOff[s];
    const int tid = threadIdx.x;
    if (ti < 0 || (ti == 0 && tj == 0)) {
        // publish L11 (k_big_trsm only kept it in LDS); nobody else reads or writes this block in this launch
        double* blk = &As[0][0]; // NB*NB doubles fit (NB * TS)
        for (int e = tid; e < NB * NB; e += WG) {
            const int k = e / NB, r = e - k * NB;
            blk[e] = (k < w && r < w && r >= k) ? F[(kb + r) + (long long)N * (kb + k)] : 0.0;
        }
        __syncthreads();
        if (tid < 64) (void)wave_potrf32(blk, NB, w, tid);
        __syncthreads();
        for (int e = tid; e < NB * NB; e += WG) {
            const int k = e / NB, r = e - k * NB;
            if (k < w && r < w && r >= k) F[(kb + r) + (long long)N * (kb + k)] = blk[e];
        }
        __syncthreads();
        if (ti < 0) return;
    }
    const int M0 = kb + w;
    const int i0 = M0 + TS * ti, j0 = M0 + TS * tj;
    for (int e = tid; e < NB * TS; e += WG) {
        const int k = e / TS, i = e - k * TS;
        const bool kin = k < w;
        As[k][i] = (kin && i0 + i < N) ? F[(i0 + i) + (long long)N * (kb + k)] : 0.0;
        Bs[k][i] = (kin && j0 + i < N) ? F[(j0 + i) + (long long)N * (kb + k)] : 0.0;
    }
    __syncthreads();
    const int ty = tid & 15, tx = tid >> 4;
    double acc[4][4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[ii][jj] = 0.0;
#pragma unroll 8
    for (int k = 0; k < NB; ++k) {
        double av[4], bv[4];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            av[ii] = As[k][4 * ty + ii];
            bv[ii] = Bs[k][4 * tx + ii];
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[ii][jj] += av[ii] * bv[jj];
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int col = j0 + 4 * tx + jj;
        if (col >= N) continue;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int row = i0 + 4 * ty + ii;
            if (row < N && row >= col) F[row + (long long)N * col] -= acc[ii][jj];
        }
    }
}

// ---- triangular solves ------------------------------------------------------------------------------------
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

// w[I] of a front: own right-hand side rows plus what the children pushed up
__device__ __forceinline__ double gather_w(const TreeView& tv, const long long* __restrict__ wOff, const double* __restrict__ wbuf,
    const double* __restrict__ yperm, int s, int nc, int I)
{
    double val = (I < nc) ? yperm[3 * tv.firstNode[s] + I] : 0.0;
    const int In = I / 3, Id = I - 3 * In;
    for (int ci = tv.childPtr[s]; ci < tv.childPtr[s + 1]; ++ci) {
        const int c = tv.child[ci];
        const int ic = tv.inv[tv.invPtr[c] + In];
        if (ic >= 0) val += wbuf[wOff[c] + frontNc(tv, c) + 3 * ic + Id];
    }
    return val;
}

__global__ __launch_bounds__(WG) void k_fwd_level(const int* __restrict__ list, TreeView tv, const long long* __restrict__ wOff,
    const double* __restrict__ fronts, double* __restrict__ wbuf, double* __restrict__ yperm)
{
    extern __shared__ double w[];
    const int s = list[blockIdx.x];
    const int N = frontN(tv, s), nc = frontNc(tv, s);
    const double* L = fronts + tv.frontOff[s];
    const int tid = threadIdx.x;
    const int col0 = 3 * tv.firstNode[s];
    for (int I = tid; I < N; I += WG) w[I] = gather_w(tv, wOff, wbuf, yperm, s, nc, I);
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

// big fronts, forward: prologue (gather) then one launch per 32-column step.
// desc = (front, first row of this chunk, 0, 0)
__global__ __launch_bounds__(WG) void k_big_fwd_gather(const int4* __restrict__ desc, TreeView tv, const long long* __restrict__ wOff,
    double* __restrict__ wbuf, const double* __restrict__ yperm)
{
    const int4 d = desc[blockIdx.x];
    const int s = d.x;
    const int N = frontN(tv, s), nc = frontNc(tv, s);
    const int I = d.y + threadIdx.x;
    if (I < N) wbuf[wOff[s] + I] = gather_w(tv, wOff, wbuf, yperm, s, nc, I);
}
// desc = (front, kb, row offset behind the pivot block, 0).  Every workgroup solves the 32x32 pivot system itself
// (the unsolved w_j stays untouched in wbuf, so there is no race); the chunk with offset 0 publishes y_j.
__global__ __launch_bounds__(WG) void k_big_fwd_step(const int4* __restrict__ desc, TreeView tv, const long long* __restrict__ wOff,
    const double* __restrict__ fronts, double* __restrict__ wbuf, double* __restrict__ yperm)
{
    __shared__ double ys[NB];
    const int4 d = desc[blockIdx.x];
    const int s = d.x, kb = d.y;
    const int N = frontN(tv, s), nc = frontNc(tv, s);
    const int wd = min(NB, nc - kb);
    const double* L = fronts + tv.frontOff[s];
    double* wv = wbuf + wOff[s];
    const int tid = threadIdx.x;
    if (tid < 64) {
        double y = (tid < wd) ? wv[kb + tid] : 0.0;
        for (int j = 0; j < wd; ++j) {
            const double Ljj = L[(kb + j) + (long long)N * (kb + j)];
            const double yj = __shfl(y, j, 64) / Ljj;
            if (tid == j) y = yj;
            else if (tid > j && tid < wd) y -= L[(kb + tid) + (long long)N * (kb + j)] * yj;
        }
        if (tid < NB) ys[tid] = (tid < wd) ? y : 0.0;
        if (d.z == 0 && tid < wd) yperm[3 * tv.firstNode[s] + kb + tid] = y;
    }
    __syncthreads();
    const int r = kb + wd + d.z + tid;
    if (r < N) {
        double acc = 0.0;
        for (int k = 0; k < wd; ++k) acc += L[r + (long long)N * (kb + k)] * ys[k];
        wv[r] -= acc;
    }
}

__global__ __launch_bounds__(WG) void k_bwd_level(const int* __restrict__ list, TreeView tv, const double* __restrict__ fronts,
    const double* __restrict__ yperm, double* __restrict__ xsol)
{
    extern __shared__ double x[];
    const int s = list[blockIdx.x];
    const int N = frontN(tv, s), nc = frontNc(tv, s);
    const double* L = fronts + tv.frontOff[s];
    const int* idx = tv.idx + tv.idxPtr[s];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col0 = 3 * tv.firstNode[s];
    for (int I = tid; I < N; I += WG) {
        const int In = I / 3;
        x[I] = (I < nc) ? yperm[col0 + I] : xsol[3 * idx[In] + (I - 3 * In)];
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
    for (int I = tid; I < nc; I += WG) xsol[col0 + I] = x[I];
}

// big fronts, backward prologue: y_c -= sum_{r >= nc} L(r, c) x_r  (x of the ancestors).  desc = (front, first column, 0, 0);
// one wave per column, lanes stride the rows.
__global__ __launch_bounds__(WG) void k_big_bwd_init(const int4* __restrict__ desc, TreeView tv, const double* __restrict__ fronts,
    double* __restrict__ yperm, const double* __restrict__ xsol)
{
    const int4 d = desc[blockIdx.x];
    const int s = d.x;
    const int N = frontN(tv, s), nc = frontNc(tv, s);
    const double* L = fronts + tv.frontOff[s];
    const int* idx = tv.idx + tv.idxPtr[s];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col0 = 3 * tv.firstNode[s];
    for (int c = d.y + wave; c < min(nc, d.y + 16); c += WG / 64) {
        double acc = 0.0;
        const double* Lc = L + (long long)N * c;
        for (int r = nc + lane; r < N; r += 64) {
            const int rn = r / 3;
            acc += Lc[r] * xsol[3 * idx[rn] + (r - 3 * rn)];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (lane == 0) yperm[col0 + c] -= acc;
    }
}
// desc = (front, kb, first column of this chunk, 0), steps run from the last block to the first.  Every workgroup
// solves L_jj^T x_j = t_j itself (t_j is read from yperm, x_j goes to xsol: no race), then updates its columns c < kb.
__global__ __launch_bounds__(WG) void k_big_bwd_step(const int4* __restrict__ desc, TreeView tv, const double* __restrict__ fronts,
    double* __restrict__ yperm, double* __restrict__ xsol)
{
    __shared__ double xs[NB];
    const int4 d = desc[blockIdx.x];
    const int s = d.x, kb = d.y;
    const int N = frontN(tv, s), nc = frontNc(tv, s);
    const int wd = min(NB, nc - kb);
    const double* L = fronts + tv.frontOff[s];
    const int col0 = 3 * tv.firstNode[s];
    const int tid = threadIdx.x;
    if (tid < 64) {
        double t = (tid < wd) ? yperm[col0 + kb + tid] : 0.0;
        for (int j = wd - 1; j >= 0; --j) {
            const double Ljj = L[(kb + j) + (long long)N * (kb + j)];
            const double xj = __shfl(t, j, 64) / Ljj;
            if (tid == j) t = xj;
            else if (tid < j) t -= L[(kb + j) + (long long)N * (kb + tid)] * xj;
        }
        if (tid < NB) xs[tid] = (tid < wd) ? t : 0.0;
        if (d.z == 0 && tid < wd) xsol[col0 + kb + tid] = t;
    }
    __syncthreads();
    const int c = d.z + tid;
    if (c < kb) {
        const double* Lc = L + (long long)N * c + kb;
        double acc = 0.0;
        for (int k = 0; k < wd; ++k) acc += Lc[k] * xs[k];
        yperm[col0 + c] -= acc;
    }
}

} // namespace

void MfNumeric::setup(const MfSymbolic& sym, hipStream_t stream)
{
    sym_ = &sym;
    stream_ = stream;
    ns_ = sym.ns;
    nLevels_ = (int)sym.levelPtr.size() - 1;
    fronts_.alloc((size_t)sym.frontOff[ns_]);
    w_.alloc((size_t)sym.wOff[ns_]);
    yperm_.alloc((size_t)sym.n);
    xsol_.alloc((size_t)sym.n);
    idx_.upload(sym.idx, stream);
    idxPtr_.upload(sym.idxPtr, stream);
    firstNode_.upload(sym.firstNode, stream);
    childPtr_.upload(sym.childPtr, stream);
    child_.upload(sym.child.empty() ? std::vector<int>{ 0 } : sym.child, stream);
    invPtr_.upload(sym.invPtr, stream);
    inv_.upload(sym.inv.empty() ? std::vector<int>{ 0 } : sym.inv, stream);
    newOf_.upload(sym.newOf, stream);
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

    int bigN = 192; // fronts wider than this go through the level-batched multi-workgroup kernels
    if (const char* e = std::getenv("IPCGPU_MF_BIGN")) bigN = std::max(NB + 1, std::min(448, std::atoi(e)));
    plan_.assign(nLevels_, LevelPlan());
    std::vector<int> smallList;
    std::vector<int2> ea;
    std::vector<int4> desc;
    size_t maxSmallLds = 0, maxSolveLds = 0;
    for (int l = 0; l < nLevels_; ++l) {
        LevelPlan& P = plan_[l];
        std::vector<int> small, big;
        for (int i = sym.levelPtr[l]; i < sym.levelPtr[l + 1]; ++i) {
            const int s = sym.levelFronts[i];
            (sym.N(s) <= bigN ? small : big).push_back(s);
        }
        // heaviest first so the tail of the level is made of short jobs
        std::sort(small.begin(), small.end(), [&](int a, int b) { return sym.N(a) > sym.N(b); });
        P.small.off = (int)smallList.size();
        P.small.cnt = (int)small.size();
        int maxN = 0;
        for (int s : small) maxN = std::max(maxN, sym.N(s));
        smallList.insert(smallList.end(), small.begin(), small.end());
        P.smallLds = (size_t)(NB * maxN + 8) * sizeof(double);
        P.solveLds = (size_t)std::max(maxN, 1) * sizeof(double);
        maxSmallLds = std::max(maxSmallLds, P.smallLds);
        maxSolveLds = std::max(maxSolveLds, P.solveLds);
        // extend-add descriptors (fronts with children only)
        P.ea.off = (int)ea.size();
        for (int i = sym.levelPtr[l]; i < sym.levelPtr[l + 1]; ++i) {
            const int s = sym.levelFronts[i];
            if (sym.childPtr[s + 1] == sym.childPtr[s]) continue;
            const long long total = (long long)sym.N(s) * sym.N(s);
            const int chunks = (int)((total + WG * EA_ITEMS - 1) / (WG * EA_ITEMS));
            for (int c = 0; c < chunks; ++c) ea.push_back(make_int2(s, c));
        }
        P.ea.cnt = (int)ea.size() - P.ea.off;
        // big-front step descriptors
        int steps = 0;
        for (int s : big) steps = std::max(steps, (sym.nc(s) + NB - 1) / NB);
        P.trsm.assign(steps, Range());
        P.syrk.assign(steps, Range());
        P.fwd.assign(steps, Range());
        P.bwd.assign(steps, Range());
        for (int j = 0; j < steps; ++j) {
            const int kb = j * NB;
            P.trsm[j].off = (int)desc.size();
            for (int s : big) {
                if (kb >= sym.nc(s)) continue;
                const int w = std::min(NB, sym.nc(s) - kb), rows = sym.N(s) - kb - w;
                for (int r0 = 0; r0 < rows; r0 += WG) desc.push_back(make_int4(s, kb, r0, 0));
            }
            P.trsm[j].cnt = (int)desc.size() - P.trsm[j].off;
            P.syrk[j].off = (int)desc.size();
            for (int s : big) {
                if (kb >= sym.nc(s)) continue;
                const int w = std::min(NB, sym.nc(s) - kb), M = sym.N(s) - kb - w;
                const int nt = (M + TS - 1) / TS;
                if (nt == 0) desc.push_back(make_int4(s, kb, -1, -1));
                for (int ti = 0; ti < nt; ++ti)
                    for (int tj = 0; tj <= ti; ++tj) desc.push_back(make_int4(s, kb, ti, tj));
            }
            P.syrk[j].cnt = (int)desc.size() - P.syrk[j].off;
            P.fwd[j].off = (int)desc.size();
            for (int s : big) {
                if (kb >= sym.nc(s)) continue;
                const int w = std::min(NB, sym.nc(s) - kb), rows = sym.N(s) - kb - w;
                for (int r0 = 0; r0 == 0 || r0 < rows; r0 += WG) desc.push_back(make_int4(s, kb, r0, 0));
            }
            P.fwd[j].cnt = (int)desc.size() - P.fwd[j].off;
            P.bwd[j].off = (int)desc.size();
            for (int s : big) {
                if (kb >= sym.nc(s)) continue;
                for (int c0 = 0; c0 == 0 || c0 < kb; c0 += WG) desc.push_back(make_int4(s, kb, c0, 0));
            }
            P.bwd[j].cnt = (int)desc.size() - P.bwd[j].off;
        }
        P.fwdGather.off = (int)desc.size();
        for (int s : big)
            for (int r0 = 0; r0 < sym.N(s); r0 += WG) desc.push_back(make_int4(s, r0, 0, 0));
        P.fwdGather.cnt = (int)desc.size() - P.fwdGather.off;
        P.bwdInit.off = (int)desc.size();
        for (int s : big)
            if (sym.N(s) > sym.nc(s))
                for (int c0 = 0; c0 < sym.nc(s); c0 += 16) desc.push_back(make_int4(s, c0, 0, 0));
        P.bwdInit.cnt = (int)desc.size() - P.bwdInit.off;
    }
    if (smallList.empty()) smallList.push_back(0);
    smallList_.upload(smallList, stream);
    if (ea.empty()) ea.push_back(make_int2(0, 0));
    eaDesc_.upload(ea.data(), ea.size(), stream);
    if (desc.empty()) desc.push_back(make_int4(0, 0, 0, 0));
    desc_.upload(desc.data(), desc.size(), stream);
    if (maxSmallLds > 64 * 1024)
        HIP_CHECK(hipFuncSetAttribute((const void*)k_factor_front, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxSmallLds));
    if (maxSolveLds > 64 * 1024) {
        HIP_CHECK(hipFuncSetAttribute((const void*)k_fwd_level, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxSolveLds));
        HIP_CHECK(hipFuncSetAttribute((const void*)k_bwd_level, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxSolveLds));
    }
    HIP_CHECK(hipStreamSynchronize(stream));
}

bool MfNumeric::factorize(const double* a_dev)
{
    if (!sym_) throw StateError("factorize before analyze_pattern");
    const MfSymbolic& sym = *sym_;
    TreeView tv{ frontOff_.p, idxPtr_.p, firstNode_.p, childPtr_.p, child_.p, invPtr_.p, inv_.p, idx_.p };
    fronts_.zero(stream_);
    flag_.zero(stream_);
    const int nnz = (int)sym.aDst.size();
    hipLaunchKernelGGL(k_scatter_a, dim3((nnz + 255) / 256), dim3(256), 0, stream_, nnz, a_dev, aDst_.p, fronts_.p);
    for (int l = 0; l < nLevels_; ++l) {
        const LevelPlan& P = plan_[l];
        if (P.ea.cnt) hipLaunchKernelGGL(k_extend_add, dim3(P.ea.cnt), dim3(WG), 0, stream_, eaDesc_.p + P.ea.off, tv, fronts_.p);
        if (P.small.cnt)
            hipLaunchKernelGGL(k_factor_front, dim3(P.small.cnt), dim3(WG), P.smallLds, stream_, smallList_.p + P.small.off, tv, fronts_.p,
                flag_.p);
        for (size_t j = 0; j < P.trsm.size(); ++j) {
            if (P.trsm[j].cnt)
                hipLaunchKernelGGL(k_big_trsm, dim3(P.trsm[j].cnt), dim3(WG), 0, stream_, desc_.p + P.trsm[j].off, tv, fronts_.p, flag_.p);
            if (P.syrk[j].cnt) hipLaunchKernelGGL(k_big_syrk, dim3(P.syrk[j].cnt), dim3(WG), 0, stream_, desc_.p + P.syrk[j].off, tv, fronts_.p);
        }
    }
    HIP_CHECK(hipMemcpyAsync(hflag_.p, flag_.p, sizeof(int), hipMemcpyDeviceToHost, stream_));
    HIP_CHECK(hipStreamSynchronize(stream_));
    return hflag_.p[0] == 0;
}

void MfNumeric::solve(const double* rhs_dev, double* x_dev)
{
    if (!sym_) throw StateError("solve before analyze_pattern");
    const MfSymbolic& sym = *sym_;
    TreeView tv{ frontOff_.p, idxPtr_.p, firstNode_.p, childPtr_.p, child_.p, invPtr_.p, inv_.p, idx_.p };
    const int n3 = sym.n;
    hipLaunchKernelGGL(k_permute_rhs, dim3((n3 + 255) / 256), dim3(256), 0, stream_, sym.nn, newOf_.p, rhs_dev, yperm_.p);
    for (int l = 0; l < nLevels_; ++l) {
        const LevelPlan& P = plan_[l];
        if (P.small.cnt)
            hipLaunchKernelGGL(k_fwd_level, dim3(P.small.cnt), dim3(WG), P.solveLds, stream_, smallList_.p + P.small.off, tv, wOff_.p,
                fronts_.p, w_.p, yperm_.p);
        if (P.fwdGather.cnt)
            hipLaunchKernelGGL(k_big_fwd_gather, dim3(P.fwdGather.cnt), dim3(WG), 0, stream_, desc_.p + P.fwdGather.off, tv, wOff_.p, w_.p,
                yperm_.p);
        for (size_t j = 0; j < P.fwd.size(); ++j)
            if (P.fwd[j].cnt)
                hipLaunchKernelGGL(k_big_fwd_step, dim3(P.fwd[j].cnt), dim3(WG), 0, stream_, desc_.p + P.fwd[j].off, tv, wOff_.p, fronts_.p,
                    w_.p, yperm_.p);
    }
    for (int l = nLevels_ - 1; l >= 0; --l) {
        const LevelPlan& P = plan_[l];
        if (P.bwdInit.cnt)
            hipLaunchKernelGGL(k_big_bwd_init, dim3(P.bwdInit.cnt), dim3(WG), 0, stream_, desc_.p + P.bwdInit.off, tv, fronts_.p, yperm_.p,
                xsol_.p);
        for (int j = (int)P.bwd.size() - 1; j >= 0; --j)
            if (P.bwd[j].cnt)
                hipLaunchKernelGGL(k_big_bwd_step, dim3(P.bwd[j].cnt), dim3(WG), 0, stream_, desc_.p + P.bwd[j].off, tv, fronts_.p, yperm_.p,
                    xsol_.p);
        if (P.small.cnt)
            hipLaunchKernelGGL(k_bwd_level, dim3(P.small.cnt), dim3(WG), P.solveLds, stream_, smallList_.p + P.small.off, tv, fronts_.p,
                yperm_.p, xsol_.p);
    }
    hipLaunchKernelGGL(k_unpermute_x, dim3((n3 + 255) / 256), dim3(256), 0, stream_, sym.nn, newOf_.p, xsol_.p, x_dev);
}

} // namespace ipcgpu
