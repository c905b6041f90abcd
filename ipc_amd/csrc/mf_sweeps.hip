// GPU multifrontal Cholesky: the triangular sweeps (forward level by level up the assembly tree, backward down it) on the factor mf_numeric.hip leaves in HBM.
// Split out of mf_numeric.hip in round 5; the kernels and their launch order are unchanged.
#include "mf_kernels.h"
#include <string>
#include <vector>
#include <algorithm>

#ifndef FWD_NARROW
#define FWD_NARROW 1 // the forward sweep of the narrow levels with two waves per workgroup as well (0: four)
#endif
#ifndef BWD_NARROW_MAX_N
#define BWD_NARROW_MAX_N 160 // levels whose fronts have at most this many rows sweep backward with two waves per workgroup (0 = never)
#endif

namespace ipcgpu {

namespace {

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

// w[I] of a front: own right-hand side rows plus what the children pushed up.  Four children at a time, level by level of
// the index chain (child -> its map and offsets -> map entry -> value): as a plain loop over the children every child paid
// its own four dependent memory round trips.  The sum runs over the children in order, as before.
__device__ __forceinline__ double gather_w(const TreeView& tv, const long long* __restrict__ wOff, const double* __restrict__ wbuf,
    const double* __restrict__ bperm, int s, int nc, int I)
{
    const int cb = tv.childPtr[s], ce = tv.childPtr[s + 1];
    double val = (I < nc) ? bperm[3 * tv.firstNode[s] + I] : 0.0;
    const int In = I / 3, Id = I - 3 * In;
    for (int c0 = cb; c0 < ce; c0 += 4) {
        int ip[4];
        long long base[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = tv.child[min(c0 + q, ce - 1)];
            ip[q] = tv.invPtr[c];
            base[q] = wOff[c] + frontNc(tv, c) + Id;
        }
        int ic[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) ic[q] = tv.inv[ip[q] + In];
        double v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = wbuf[base[q] + 3 * max(ic[q], 0)];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (c0 + q < ce && ic[q] >= 0) val += v[q];
    }
    return val;
}

// forward sweep over the nc x nc triangle of one front held in LDS (w1[0..nc)): y_b = Inv_b w_b, then the rows below
// Both sweeps are software-pipelined: the inverse block and the L entries of a thread's first row / column for step b + 1
// are requested before the barriers of step b, so their HBM / L2 latency overlaps the (short) compute of the step.
template <int NT>
__device__ __forceinline__ void fwd_triangle(const double* __restrict__ L, int N, int nc, int rowEnd, const double* __restrict__ dblk,
    double* w1, int tid)
{
    double dn[NB], ln[NB]; // prefetched: column `tid`-th row of the next inverse block; L(i, kb..kb+31) of this thread's first row
    auto fetch = [&](int kb, const double* blk) {
        const int wd = min(NB, nc - kb);
        if (tid < NB) {
#pragma unroll
            for (int c = 0; c < NB; ++c) dn[c] = blk[c * NB + tid];
        }
        const int i = kb + wd + tid;
        if (i < rowEnd) {
#pragma unroll
            for (int k = 0; k < NB; ++k) ln[k] = (k < wd) ? L[i + (long long)N * (kb + k)] : 0.0;
        }
    };
    if (nc > 0) fetch(0, dblk);
    for (int kb = 0; kb < nc; kb += NB, dblk += NB * NB) {
        const int wd = min(NB, nc - kb);
        double y = 0.0;
        if (tid < NB) { // four partial sums: a single accumulator is a 32-long dependent chain
            double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
            for (int c = 0; c < NB; c += 4) {
                p0 += dn[c] * ((c < wd) ? w1[kb + c] : 0.0);
                p1 += dn[c + 1] * ((c + 1 < wd) ? w1[kb + c + 1] : 0.0);
                p2 += dn[c + 2] * ((c + 2 < wd) ? w1[kb + c + 2] : 0.0);
                p3 += dn[c + 3] * ((c + 3 < wd) ? w1[kb + c + 3] : 0.0);
            }
            y = (p0 + p1) + (p2 + p3);
        }
        __syncthreads();
        if (tid < wd) w1[kb + tid] = y;
        __syncthreads();
        const int i0 = kb + wd + tid;
        double acc0 = 0.0;
        if (i0 < rowEnd) {
            double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
            for (int k = 0; k < NB; k += 4) {
                p0 += ln[k] * ((k < wd) ? w1[kb + k] : 0.0);
                p1 += ln[k + 1] * ((k + 1 < wd) ? w1[kb + k + 1] : 0.0);
                p2 += ln[k + 2] * ((k + 2 < wd) ? w1[kb + k + 2] : 0.0);
                p3 += ln[k + 3] * ((k + 3 < wd) ? w1[kb + k + 3] : 0.0);
            }
            acc0 = (p0 + p1) + (p2 + p3);
        }
        if (kb + NB < nc) fetch(kb + NB, dblk + NB * NB); // in flight across the barrier below and the next block solve
        if (i0 < rowEnd) w1[i0] -= acc0;
        for (int i = i0 + NT; i < rowEnd; i += NT) {
            double acc = 0.0;
#pragma unroll 8
            for (int k = 0; k < wd; ++k) acc += L[i + (long long)N * (kb + k)] * w1[kb + k];
            w1[i] -= acc;
        }
        __syncthreads();
    }
}

// backward sweep: t[0..nc) holds y - L21^T x2 on entry, x1 on exit.  x_b = Inv_b^T t_b, then the columns to the left
template <int NT>
__device__ __forceinline__ void bwd_triangle(const double* __restrict__ L, int N, int nc, const double* __restrict__ dblk0, double* t,
    double* invs /* NB * LDP */, int tid)
{
    const int nblk = (nc + NB - 1) / NB;
    for (int b = nblk - 1; b >= 0; --b) {
        const int kb = b * NB;
        const int wd = min(NB, nc - kb);
        const double* dblk = dblk0 + (long long)b * (NB * NB);
        for (int e = tid; e < NB * NB; e += NT) invs[(e >> 5) * LDP + (e & 31)] = dblk[e];
        __syncthreads();
        double x = 0.0;
        if (tid < NB) {
            double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
            for (int r = 0; r < NB; r += 4) {
                p0 += invs[tid * LDP + r] * ((r < wd) ? t[kb + r] : 0.0);
                p1 += invs[tid * LDP + r + 1] * ((r + 1 < wd) ? t[kb + r + 1] : 0.0);
                p2 += invs[tid * LDP + r + 2] * ((r + 2 < wd) ? t[kb + r + 2] : 0.0);
                p3 += invs[tid * LDP + r + 3] * ((r + 3 < wd) ? t[kb + r + 3] : 0.0);
            }
            x = (p0 + p1) + (p2 + p3);
        }
        __syncthreads();
        if (tid < wd) t[kb + tid] = x;
        __syncthreads();
        for (int c = tid; c < kb; c += NT) {
            const double* Lc = L + (long long)N * c + kb;
            double p0 = 0.0, p1 = 0.0;
#pragma unroll 8
            for (int k = 0; k + 1 < wd; k += 2) {
                p0 += Lc[k] * t[kb + k];
                p1 += Lc[k + 1] * t[kb + k + 1];
            }
            if (wd & 1) p0 += Lc[wd - 1] * t[kb + wd - 1];
            t[c] -= p0 + p1;
        }
        __syncthreads();
    }
}

// (NT = 256, or 128 on levels of narrow fronts: see k_bwd_level)
template <int NT>
__global__ __launch_bounds__(NT) void k_fwd_level(const int* __restrict__ list, TreeView tv, const long long* __restrict__ wOff,
    const double* __restrict__ fronts, const double* __restrict__ dinv, double* __restrict__ wbuf, const double* __restrict__ bperm,
    double* __restrict__ yperm)
{
    extern __shared__ double w[];
    const int s = list[blockIdx.x];
    const int N = frontN(tv, s), nc = frontNc(tv, s);
    const double* L = fronts + tv.frontOff[s];
    const int tid = threadIdx.x;
    const int col0 = 3 * tv.firstNode[s];
    for (int I = tid; I < N; I += NT) w[I] = gather_w(tv, wOff, wbuf, bperm, s, nc, I);
    __syncthreads();
    fwd_triangle<NT>(L, N, nc, N, dinv + tv.dinvOff[s] * (NB * NB), w, tid);
    double* wo = wbuf + wOff[s];
    for (int I = tid; I < N; I += NT) {
        wo[I] = w[I]; // rows >= nc carry (children contributions - L21 y) up to the parent
        if (I < nc) yperm[col0 + I] = w[I];
    }
}

// big fronts, forward: one workgroup sweeps the triangle ...
__global__ __launch_bounds__(WGT) void k_big_fwd_tri(const int* __restrict__ list, TreeView tv, const long long* __restrict__ wOff,
    const double* __restrict__ fronts, const double* __restrict__ dinv, const double* __restrict__ wbuf, const double* __restrict__ bperm,
    double* __restrict__ yperm)
{
    extern __shared__ double w[];
    const int s = list[blockIdx.x];
    const int N = frontN(tv, s), nc = frontNc(tv, s);
    const double* L = fronts + tv.frontOff[s];
    const int tid = threadIdx.x;
    const int col0 = 3 * tv.firstNode[s];
    for (int I = tid; I < nc; I += WGT) w[I] = gather_w(tv, wOff, wbuf, bperm, s, nc, I);
    __syncthreads();
    fwd_triangle<WGT>(L, N, nc, nc, dinv + tv.dinvOff[s] * (NB * NB), w, tid);
    for (int I = tid; I < nc; I += WGT) yperm[col0 + I] = w[I];
}
// ... then the rectangle below it: w2[r] = (children) - sum_c L(r, c) y_c.  desc = (front, first row behind nc, 0, 0);
// 32 rows per workgroup, eight column groups
__global__ __launch_bounds__(WG) void k_big_fwd_rect(const int4* __restrict__ desc, TreeView tv, const long long* __restrict__ wOff,
    const double* __restrict__ fronts, double* __restrict__ wbuf, const double* __restrict__ yperm)
{
    __shared__ double part[WG];
    const int4 d = desc[blockIdx.x];
    const int s = d.x;
    const int N = frontN(tv, s), nc = frontNc(tv, s);
    const double* L = fronts + tv.frontOff[s];
    const double* y = yperm + 3 * tv.firstNode[s];
    // 32 rows x 8 column groups per workgroup (64 x 4 before): the products of a row are a chain of dependent load rounds, 16
    // loads each, and a top-level front has few rows -- more, shorter chains on more CUs
    const int lane = threadIdx.x & (MV_ROWS - 1), cg = threadIdx.x / MV_ROWS;
    constexpr int NCG = WG / MV_ROWS;
    const int r = nc + d.y + lane;
    // what the children pushed up for this row: requested first, its index chain resolves while the products run
    const double up = (cg == 0 && r < N) ? gather_w(tv, wOff, wbuf, yperm, s, nc, r) : 0.0;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    {
        const double* Lr = L + min(r, N - 1); // rows past the end are clamped and dropped
        const int per = (nc + NCG - 1) / NCG;
        const int c0 = cg * per, c1 = min(nc, c0 + per);
        int c = c0;
        for (; c + 16 <= c1; c += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = Lr[(long long)N * (c + u)];
#pragma unroll
            for (int u = 0; u < 16; u += 4) {
                a0 += v[u] * y[c + u];
                a1 += v[u + 1] * y[c + u + 1];
                a2 += v[u + 2] * y[c + u + 2];
                a3 += v[u + 3] * y[c + u + 3];
            }
        }
        for (; c + 4 <= c1; c += 4) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = Lr[(long long)N * (c + u)];
            a0 += v[0] * y[c];
            a1 += v[1] * y[c + 1];
            a2 += v[2] * y[c + 2];
            a3 += v[3] * y[c + 3];
        }
        for (; c < c1; ++c) a0 += Lr[(long long)N * c] * y[c];
    }
    part[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (cg == 0 && r < N) {
        double tot = 0.0;
#pragma unroll
        for (int g = 0; g < NCG; ++g) tot += part[g * MV_ROWS + lane];
        wbuf[wOff[s] + r] = up - tot;
    }
}

// NT = 256, or 128 on levels of narrow fronts (round 6: a leaf front has ~90 rows; with two waves per workgroup twice as many of the 2 176 leaves of mat150 are resident and
// the level -- one latency-bound round of workgroups -- takes 22 instead of 36 us)
template <int NT>
__global__ __launch_bounds__(NT) void k_bwd_level(const int* __restrict__ list, TreeView tv, const double* __restrict__ fronts,
    const double* __restrict__ dinv, const double* __restrict__ yperm, double* __restrict__ xsol)
{
    constexpr int NW = NT / 64;
    extern __shared__ double x[];
    __shared__ double invs[NB * LDP];
    const int s = list[blockIdx.x];
    const int N = frontN(tv, s), nc = frontNc(tv, s);
    const double* L = fronts + tv.frontOff[s];
    const int* idx = tv.idx + tv.idxPtr[s];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col0 = 3 * tv.firstNode[s];
    for (int I = tid; I < N; I += NT) {
        const int In = I / 3;
        x[I] = (I < nc) ? yperm[col0 + I] : xsol[3 * idx[In] + (I - 3 * In)];
    }
    __syncthreads();
    // t = y1 - L21^T x2: a wave takes four columns at a time, lanes stride the rows (two 64-row strips per round: eight loads
    // in flight; one column and one strip at a time this loop was nc / 4 x (N - nc) / 64 dependent round trips per wave)
    {
        const int m = N - nc;
        for (int c0 = wave; c0 < nc; c0 += 4 * NW) {
            double acc[4] = { 0.0, 0.0, 0.0, 0.0 };
            for (int i0 = 0; i0 < m; i0 += 128) {
                const int ia = i0 + lane, ib = i0 + 64 + lane;
                const double xa = (ia < m) ? x[nc + min(ia, m - 1)] : 0.0, xb = (ib < m) ? x[nc + min(ib, m - 1)] : 0.0;
                double va[4], vb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double* Lc = L + (long long)N * min(c0 + NW * q, nc - 1) + nc;
                    va[q] = Lc[min(ia, m - 1)];
                    vb[q] = Lc[min(ib, m - 1)];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] += va[q] * xa + vb[q] * xb;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double t = acc[q];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
                if (lane == 0 && c0 + NW * q < nc) x[c0 + NW * q] -= t;
            }
        }
    }
    __syncthreads();
    bwd_triangle<NT>(L, N, nc, dinv + tv.dinvOff[s] * (NB * NB), x, invs, tid);
    for (int I = tid; I < nc; I += NT) xsol[col0 + I] = x[I];
}

// big fronts, backward prologue: y_c -= sum_{r >= nc} L(r, c) x_r  (x of the ancestors).  desc = (front, first column, 0, 0);
// one wave per column, lanes stride the rows.
__global__ __launch_bounds__(WG) void k_big_bwd_init(const int4* __restrict__ desc, TreeView tv, const double* __restrict__ fronts,
    double* __restrict__ yperm, const double* __restrict__ xsol)
{
    extern __shared__ double x2[]; // x of the ancestors in this front's row order, gathered once per workgroup
    const int4 d = desc[blockIdx.x];
    const int s = d.x;
    const int N = frontN(tv, s), nc = frontNc(tv, s);
    const double* L = fronts + tv.frontOff[s];
    const int* idx = tv.idx + tv.idxPtr[s];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col0 = 3 * tv.firstNode[s];
    // the four columns of a wave go together, two 64-row strips each per round: eight loads in flight instead of two, and the
    // old value of y is requested up front rather than behind the reduction
    const int m = N - nc;
    int cq[4];
    double yold[4], acc[4] = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        cq[q] = d.y + wave + 4 * q;
        yold[q] = yperm[col0 + min(cq[q], nc - 1)];
    }
    for (int r0 = nc; r0 < N; r0 += 4 * WG) { // x of the ancestors: four index loads, then four value loads, per round
        int rq[4], id[4];
        double xv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            rq[q] = min(r0 + q * WG + (int)threadIdx.x, N - 1);
            id[q] = idx[rq[q] / 3];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) xv[q] = xsol[3 * id[q] + rq[q] % 3];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (r0 + q * WG + (int)threadIdx.x < N) x2[rq[q] - nc] = xv[q];
    }
    __syncthreads();
    for (int r0 = 0; r0 < m; r0 += 128) {
        const int ra = r0 + lane, rb = r0 + 64 + lane;
        const double xa = (ra < m) ? x2[min(ra, m - 1)] : 0.0, xb = (rb < m) ? x2[min(rb, m - 1)] : 0.0;
        double va[4], vb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double* Lc = L + (long long)N * min(cq[q], nc - 1) + nc;
            va[q] = Lc[min(ra, m - 1)];
            vb[q] = Lc[min(rb, m - 1)];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += va[q] * xa + vb[q] * xb;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double t = acc[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        if (lane == 0 && cq[q] < min(nc, d.y + 16)) yperm[col0 + cq[q]] = yold[q] - t;
    }

}
// ... then one workgroup sweeps the transposed triangle
__global__ __launch_bounds__(WGT) void k_big_bwd_tri(const int* __restrict__ list, TreeView tv, const double* __restrict__ fronts,
    const double* __restrict__ dinv, const double* __restrict__ yperm, double* __restrict__ xsol)
{
    extern __shared__ double t[];
    __shared__ double invs[NB * LDP];
    const int s = list[blockIdx.x];
    const int N = frontN(tv, s), nc = frontNc(tv, s);
    const double* L = fronts + tv.frontOff[s];
    const int tid = threadIdx.x;
    const int col0 = 3 * tv.firstNode[s];
    for (int I = tid; I < nc; I += WGT) t[I] = yperm[col0 + I];
    __syncthreads();
    bwd_triangle<WGT>(L, N, nc, dinv + tv.dinvOff[s] * (NB * NB), t, invs, tid);
    for (int I = tid; I < nc; I += WGT) xsol[col0 + I] = t[I];
}

// forward: y1 = X w1.  desc = (front, first row, 0, 0): 32 rows per workgroup, eight column groups.
__global__ __launch_bounds__(WG) void k_xinv_fwd(const int4* __restrict__ desc, TreeView tv, XinvView xv, const long long* __restrict__ wOff,
    const double* __restrict__ wbuf, const double* __restrict__ bperm, double* __restrict__ yperm)
{
    extern __shared__ double w1[];
    __shared__ double part[WG];
    const int4 d = desc[blockIdx.x];
    const int s = d.x, r0 = d.y;
    const int nc = frontNc(tv, s);
    const double* X = xv.X + xv.xOff[s];
    const int tid = threadIdx.x, lane = tid & (MV_ROWS - 1), cg = tid / MV_ROWS; // 32 rows x 8 column groups (see k_big_fwd_rect)
    constexpr int NCG = WG / MV_ROWS;
    const int cols = min(nc, r0 + MV_ROWS); // X(r, c) = 0 for c > r
    for (int I = tid; I < cols; I += WG) w1[I] = gather_w(tv, wOff, wbuf, bperm, s, nc, I);
    __syncthreads();
    const int r = r0 + lane;
    double acc0 = 0.0, acc1 = 0.0;
    if (r < nc) {
        const int per = (cols + NCG - 1) / NCG, cb = cg * per, ce = min(min(cols, cb + per), r + 1);
        const double* Xr = X + r;
        int c = cb;
        for (; c + 15 < ce; c += 16) { // sixteen loads in flight per lane
            double x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) x[u] = Xr[(long long)nc * (c + u)];
#pragma unroll
            for (int u = 0; u < 16; u += 2) {
                acc0 += x[u] * w1[c + u];
                acc1 += x[u + 1] * w1[c + u + 1];
            }
        }
        for (; c + 3 < ce; c += 4) {
            double x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = Xr[(long long)nc * (c + u)];
#pragma unroll
            for (int u = 0; u < 4; u += 2) {
                acc0 += x[u] * w1[c + u];
                acc1 += x[u + 1] * w1[c + u + 1];
            }
        }
        for (; c < ce; ++c) acc0 += Xr[(long long)nc * c] * w1[c];
    }
    part[tid] = acc0 + acc1;
    __syncthreads();
    if (cg == 0 && r < nc) {
        double tot = 0.0;
#pragma unroll
        for (int g = 0; g < NCG; ++g) tot += part[g * MV_ROWS + lane];
        yperm[3 * tv.firstNode[s] + r] = tot;
    }
}

// backward: x1 = X^T t with t = y1 - L21^T x2 (left in yperm by k_big_bwd_init).  desc = (front, first column, 0, 0):
// one wave per column (contiguous reads), 16 columns per workgroup.
__global__ __launch_bounds__(WG) void k_xinv_bwd(const int4* __restrict__ desc, TreeView tv, XinvView xv, const double* __restrict__ yperm,
    double* __restrict__ xsol)
{
    extern __shared__ double tt[];
    const int4 d = desc[blockIdx.x];
    const int s = d.x, c0 = d.y;
    const int nc = frontNc(tv, s);
    const double* X = xv.X + xv.xOff[s];
    const int col0 = 3 * tv.firstNode[s];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = c0 + threadIdx.x; r < nc; r += WG) tt[r - c0] = yperm[col0 + r];
    __syncthreads();
    // the four columns of a wave together, two 64-row strips per round (rows above a column's diagonal are zeros of X and are
    // masked): eight loads in flight instead of two
    int cq[4];
    double acc[4] = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
    for (int q = 0; q < 4; ++q) cq[q] = c0 + wave + 4 * q;
    for (int r0 = c0 + wave; r0 < nc; r0 += 128) {
        const int ra = r0 + lane, rb = r0 + 64 + lane;
        const double ta = (ra < nc) ? tt[min(ra, nc - 1) - c0] : 0.0, tb = (rb < nc) ? tt[min(rb, nc - 1) - c0] : 0.0;
        double va[4], vb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double* Xc = X + (long long)nc * min(cq[q], nc - 1);
            va[q] = Xc[min(ra, nc - 1)];
            vb[q] = Xc[min(rb, nc - 1)];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += ((ra >= cq[q]) ? va[q] * ta : 0.0) + ((rb >= cq[q]) ? vb[q] * tb : 0.0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double t = acc[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        if (lane == 0 && cq[q] < min(nc, c0 + 16)) xsol[col0 + cq[q]] = t;
    }
}

} // namespace

// the dynamic-LDS limits of the sweep kernels (called by setup(), which knows the widest front of every kind)
void MfNumeric::configureSweepKernels(size_t maxSolveLds, size_t maxBwdLds, size_t maxTriLds)
{
    if (xinvLds_ > 48 * 1024) {
        HIP_CHECK(hipFuncSetAttribute((const void*)k_xinv_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)xinvLds_));
        HIP_CHECK(hipFuncSetAttribute((const void*)k_xinv_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)xinvLds_));
    }
    if (maxSolveLds > 48 * 1024) {
        HIP_CHECK(hipFuncSetAttribute((const void*)k_fwd_level<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxSolveLds));
        HIP_CHECK(hipFuncSetAttribute((const void*)k_fwd_level<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxSolveLds));
        HIP_CHECK(hipFuncSetAttribute((const void*)k_bwd_level<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxSolveLds));
        HIP_CHECK(hipFuncSetAttribute((const void*)k_bwd_level<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxSolveLds));
    }
    if (maxBwdLds > 48 * 1024)
        HIP_CHECK(hipFuncSetAttribute((const void*)k_big_bwd_init, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxBwdLds));
    if (maxTriLds > 48 * 1024) {
        if (maxTriLds > 150 * 1024) throw StateError("a separator front is too wide for the single-workgroup triangular sweep");
        HIP_CHECK(hipFuncSetAttribute((const void*)k_big_fwd_tri, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxTriLds));
        HIP_CHECK(hipFuncSetAttribute((const void*)k_big_bwd_tri, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxTriLds));
    }
}

// the right-hand side into the elimination order (on the forward stream when the forward sweep runs beside the factorisation)
void MfNumeric::enqueuePermuteRhs(const double* rhs_dev, hipStream_t st)
{
    const int n3 = sym_->n;
    hipLaunchKernelGGL(k_permute_rhs, dim3((n3 + 255) / 256), dim3(256), 0, st, sym_->nn, newOf_.p, rhs_dev, bperm_.p);
}

void MfNumeric::solve(const double* rhs_dev, double* x_dev)
{
    if (!sym_) throw StateError("solve before analyze_pattern");
    enqueueSolve(rhs_dev, x_dev);
}

void MfNumeric::enqueueSolve(const double* rhs_dev, double* x_dev)
{
    // the permuted right-hand side stays in its own buffer: the forward kernels of a level write y into yperm while other
    // workgroups of the same launch still gather right-hand-side entries
    enqueuePermuteRhs(rhs_dev, stream_);
    for (int l = 0; l < nLevels_; ++l) {
        const LevelPlan& P = plan_[l];
        if (P.xinvFwd.cnt && sidePending_) HIP_CHECK(hipStreamWaitEvent(stream_, evInvDone_[l], 0));
        enqueueForwardLevel(l, stream_);
        if (world_ > 1) exchangeUpdateVectors(l); // update vectors of this level's fronts -> the rank that executes their parent (mf_exchange.hip)
    }
    enqueueBackward(x_dev);
}

void MfNumeric::enqueueForwardLevel(int l, hipStream_t st)
{
    TreeView tv{ frontOff_.p, idxPtr_.p, firstNode_.p, childPtr_.p, child_.p, invPtr_.p, inv_.p, idx_.p, dinvOff_.p };
    XinvView xv{ xinvOff_.p, xinvX_.p, xinvT_.p };
    const LevelPlan& P = plan_[l];
    if (P.small.cnt)
    {
        if (FWD_NARROW && P.solveLds <= BWD_NARROW_MAX_N * sizeof(double))
            hipLaunchKernelGGL(k_fwd_level<128>, dim3(P.small.cnt), dim3(128), P.solveLds, st, smallList_.p + P.small.off, tv, wOff_.p, fronts_.p, dinv_.p, w_.p,
                bperm_.p, yperm_.p);
        else
            hipLaunchKernelGGL(k_fwd_level<256>, dim3(P.small.cnt), dim3(WG), P.solveLds, st, smallList_.p + P.small.off, tv, wOff_.p, fronts_.p, dinv_.p, w_.p,
                bperm_.p, yperm_.p);
    }
    if (P.bigTri.cnt)
        hipLaunchKernelGGL(k_big_fwd_tri, dim3(P.bigTri.cnt), dim3(WGT), P.triLds, st, triList_.p + P.bigTri.off, tv, wOff_.p, fronts_.p, dinv_.p, w_.p,
            bperm_.p, yperm_.p);
    if (P.xinvFwd.cnt)
        hipLaunchKernelGGL(k_xinv_fwd, dim3(P.xinvFwd.cnt), dim3(WG), xinvLds_, st, xinvDesc_.p + P.xinvFwd.off, tv, xv, wOff_.p, w_.p, bperm_.p, yperm_.p);
    if (P.fwdRect.cnt)
        hipLaunchKernelGGL(k_big_fwd_rect, dim3(P.fwdRect.cnt), dim3(WG), 0, st, desc_.p + P.fwdRect.off, tv, wOff_.p, fronts_.p, w_.p, yperm_.p);
}

void MfNumeric::enqueueBackward(double* x_dev)
{
    const MfSymbolic& sym = *sym_;
    TreeView tv{ frontOff_.p, idxPtr_.p, firstNode_.p, childPtr_.p, child_.p, invPtr_.p, inv_.p, idx_.p, dinvOff_.p };
    XinvView xv{ xinvOff_.p, xinvX_.p, xinvT_.p };
    const int n3 = sym.n;
#ifdef MF_BWD_PROBE // diagnosis build (-DMF_BWD_PROBE): timing events between the launches of the backward sweep, printed once (profiles/r06_backward_sweep_probe.txt)
    static int probeCall = 0;
    const bool probe = ++probeCall == 40;
    std::vector<hipEvent_t> pe;
    std::vector<std::string> pn;
    auto mark = [&](const char* what, int l) {
        if (!probe) return;
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        HIP_CHECK(hipEventRecord(e, stream_));
        pe.push_back(e);
        pn.push_back(std::string(what) + " L" + std::to_string(l));
    };
    mark("start", nLevels_);
#else
    auto mark = [](const char*, int) {};
#endif
    for (int l = nLevels_ - 1; l >= 0; --l) {
        const LevelPlan& P = plan_[l];
        if (P.bwdInit.cnt)
            hipLaunchKernelGGL(k_big_bwd_init, dim3(P.bwdInit.cnt), dim3(WG), P.bwdLds, stream_, desc_.p + P.bwdInit.off, tv, fronts_.p, yperm_.p,
                xsol_.p);
        if (P.bwdInit.cnt) mark("bwd_init", l);
        if (P.bigTri.cnt)
            hipLaunchKernelGGL(k_big_bwd_tri, dim3(P.bigTri.cnt), dim3(WGT), P.triLds, stream_, triList_.p + P.bigTri.off, tv, fronts_.p,
                dinv_.p, yperm_.p, xsol_.p);
        if (P.bigTri.cnt) mark("bwd_tri", l);
        if (P.xinvBwd.cnt)
            hipLaunchKernelGGL(k_xinv_bwd, dim3(P.xinvBwd.cnt), dim3(WG), xinvLds_, stream_, xinvDesc_.p + P.xinvBwd.off, tv, xv, yperm_.p,
                xsol_.p);
        if (P.small.cnt) {
            if (P.solveLds <= BWD_NARROW_MAX_N * sizeof(double))
                hipLaunchKernelGGL(k_bwd_level<128>, dim3(P.small.cnt), dim3(128), P.solveLds, stream_, smallList_.p + P.small.off, tv, fronts_.p, dinv_.p,
                    yperm_.p, xsol_.p);
            else
                hipLaunchKernelGGL(k_bwd_level<256>, dim3(P.small.cnt), dim3(WG), P.solveLds, stream_, smallList_.p + P.small.off, tv, fronts_.p, dinv_.p,
                    yperm_.p, xsol_.p);
        }
        if (P.xinvBwd.cnt || P.small.cnt) mark("xinv/small", l);
        if (world_ > 1) exchange(xchg_[l].opsX); // solution entries of this level's fronts above the cut -> the ranks that execute fronts below them
    }
#ifdef MF_BWD_PROBE
    if (probe) {
        HIP_CHECK(hipStreamSynchronize(stream_));
        for (size_t i = 1; i < pe.size(); ++i) {
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, pe[i - 1], pe[i]));
            fprintf(stderr, "bwd probe: %-16s %7.1f us\n", pn[i].c_str(), 1e3 * ms);
        }
    }
#endif
    if (world_ > 1) reduceSolution(); // every rank holds the solution of the fronts it executed: sum of the masked parts (mf_exchange.hip)
    hipLaunchKernelGGL(k_unpermute_x, dim3((n3 + 255) / 256), dim3(256), 0, stream_, sym.nn, newOf_.p, xsol_.p, x_dev);
}

} // namespace ipcgpu
