// GPU multifrontal Cholesky, numeric phase (gfx950, fp64).
//
// Data layout in HBM: every front is a full N x N column-major square (ld = N) inside one buffer; the leading
// nc columns become the factor panel [L11; L21], the trailing (N-nc)^2 block is the update matrix that the
// parent gathers.  All fronts stay resident (sized for 288 GB of HBM3E: ~0.5 GB for a 45 K-node sheet), so
// there is no stack management and a child update is read in place.  Next to the fronts lives `dinv`: the explicit
// inverse of every 32 x 32 diagonal block of L (column-major, 8 KB each).  The triangular solves and the panel TRSMs
// multiply by these inverses instead of substituting, which turns 32-long dependent chains (one global load, one
// division per link) into independent FMAs.
//
// Scheduling: the assembly tree is processed level by level; inside a level fronts are independent.
//   extend-add   gather formulation (each parent entry sums its children through inverse index maps):
//                race-free and bit-reproducible, no atomics
//   small fronts one 256-thread workgroup per front: 32-column panels staged in LDS, wave-level Cholesky of the
//                32 x 32 pivot block (cross-lane traffic through v_readlane, the pivot index is a compile-time
//                constant), inverse of the pivot block, TRSM as a product with it, 4x4 register tiles for the update
//   big fronts   level-batched 32-column steps, ONE launch per step with look-ahead: while 64x64 tiles apply panel j
//                to the trailing matrix (role A), other workgroups (role B) apply panel j to their rows of panel
//                j+1, factor its pivot block (redundantly per workgroup, cheaper than a launch boundary) and solve
//                their rows.  The factored pivot block itself is never written: only its inverse is needed later.
//   solve        small fronts: one workgroup per front, vectors in LDS.  Big fronts: the nc x nc triangle is swept
//                by one 512-thread workgroup per front (its only sequential part), the (N-nc) x nc rectangle is a
//                row-parallel matrix-vector product in a second launch.
// No vendor BLAS is involved: rocSOLVER's potrf / rocBLAS' trsm+syrk cost ~150 tiny launches per front.
#include <climits>
#include "mf_kernels.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace ipcgpu {

namespace {

__global__ void k_publish_flag(const int* __restrict__ flag, int* __restrict__ mapped) { mapped[0] = flag[0]; }

// values of A in fused-front order: the fused kernel then reads its entries contiguously instead of chasing a[aSrc[e]]
// (the first kernel of every factorisation: it also clears the pivot flag -- a memset of four bytes was a launch of its own on the critical path)
__global__ void k_gather_a(int cnt, const int* __restrict__ src, const double* __restrict__ a, double* __restrict__ aP, int* __restrict__ flag)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) flag[0] = 0;
    if (k < cnt) aP[k] = a[src[k]];
}


// desc = (record, ti, tj, 0): one 64 x 64 tile (ti >= tj) of a parent front.  `record` indexes a packed 64-int descriptor
// (layout of the fused kernel: [0,1] front offset [2] N [8] #children in this record [9] next record or -1; child q at
// 16 + 6 q: [0,1] front offset [2] N [3] nc [4] inverse-map offset) so that front, child list, child geometry and map
// offsets arrive in one load instead of five dependent ones.  Per child the parent-row / parent-column -> child-index maps of
// the tile are built once in LDS; lanes run along rows, which are (mostly) consecutive in the child as well, the four waves
// split the 64 columns.  The gathers are unconditional (clamped address, value selected afterwards): all in flight together.
// ownOnly: the level's Schur kernel gathers the children for the update block itself (k_big_schur64_ea): only columns < nc are written here
// ---- entries of A -> slots of the fronts, and their lists, on the device (round 5) --------------------------------------------------------------------
// A pattern change of a contact scene used to spend 2.9 ms on the host computing, for every entry of the user's CSR matrix, the front that owns it and its
// offset in that front (mf_entry_destinations, 16 threads) and 3.0 ms sorting the entries by destination (a counting sort into pinned staging buffers, then
// four uploads).  Both are a few launches here, on the pattern that is in HBM anyway: k_entry_dst (a wave per row: permuted indices, owning front, row by
// binary search in the front's sorted index list; bucket = the front for a single-workgroup front, the extend-add tile for a batched one; histogram),
// k_scan_exclusive, k_entry_scatter (slot = bucket start + ticket).  The order inside a bucket is whatever the tickets give: every entry has a slot of its
// own, so the assembled fronts do not depend on it.  tests/test_gpu_parity.py pins k_entry_dst on the host function, bit for bit.
struct EntryView {
    const int *ia, *ja;
    int nRows, ns;
    const int *newOf, *nodeFront, *firstNode, *idxPtr, *idx;
    const long long* frontOff;
    const int4* frontInfo; // per front: (kind: -1 another rank's, 0 single-workgroup, 1 batched; first extend-add tile; tile columns kept per tile row; 0)
};
__global__ __launch_bounds__(256) void k_entry_dst(EntryView v, long long* __restrict__ dst, int* __restrict__ bucket, int* __restrict__ hist)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= v.nRows) return;
    const int pr = 3 * v.newOf[r / 3] + r % 3;
    for (int k = v.ia[r] + lane; k < v.ia[r + 1]; k += 64) {
        const int c = v.ja[k];
        const int pc = 3 * v.newOf[c / 3] + c % 3;
        const int i = max(pr, pc), j = min(pr, pc); // lower triangle of the permuted matrix: row i, column j
        const int s = v.nodeFront[j / 3];
        const int f = v.firstNode[s], l = v.firstNode[s + 1];
        const int nIdx = v.idxPtr[s + 1] - v.idxPtr[s];
        int lr;
        if (i / 3 < l) lr = i - 3 * f;
        else { // a row of the structure behind the own nodes: position of node i / 3 in the front's sorted index list
            const int* b = v.idx + v.idxPtr[s] + (l - f);
            int lo = 0, hi = nIdx - (l - f);
            const int key = i / 3;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (b[mid] < key) lo = mid + 1;
                else hi = mid;
            }
            lr = 3 * ((l - f) + lo) + i % 3;
        }
        const int lc = j - 3 * f;
        dst[k] = v.frontOff[s] + lr + (long long)(3 * nIdx) * lc;
        const int4 info = v.frontInfo[s];
        int bkt = -1;
        if (info.x == 0) bkt = s;
        else if (info.x > 0) {
            const int ti = lr / TS, tj = lc / TS, cT = info.z;
            bkt = v.ns + info.y + (ti <= cT ? ti * (ti + 1) / 2 : cT * (cT + 1) / 2 + (ti - cT) * cT) + tj;
        }
        bucket[k] = bkt;
        if (bkt >= 0) atomicAdd(&hist[bkt], 1);
    }
}
// exclusive prefix sums of n counts by ONE workgroup of 1024 (n is a few hundred thousand at most); out[n] = the total
__global__ __launch_bounds__(1024) void k_scan_exclusive(int n, const int* __restrict__ in, int* __restrict__ out)
{
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int chunk = (n + 1023) / 1024;
    const int b = min(t * chunk, n), e = min(b + chunk, n);
    int sum = 0;
    for (int i = b; i < e; ++i) sum += in[i];
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) { // Hillis-Steele over the 1024 partial sums
        const int x = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    int run = part[t] - sum; // exclusive prefix of this thread's chunk
    for (int i = b; i < e; ++i) {
        out[i] = run;
        run += in[i];
    }
    if (t == 1023) out[n] = part[1023];
}
__global__ void k_entry_scatter(int nnz, int ns, const long long* __restrict__ dst, const int* __restrict__ bucket, const int* __restrict__ start,
    int* __restrict__ cursor, const long long* __restrict__ frontOff, int* __restrict__ aSrc, int* __restrict__ aLoc, int* __restrict__ bigSrc,
    long long* __restrict__ bigDst)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    const int b = bucket[k];
    if (b < 0) return;
    const int pos = start[b] + atomicAdd(&cursor[b], 1);
    if (b < ns) {
        aSrc[pos] = k;
        aLoc[pos] = (int)(dst[k] - frontOff[b]); // row + N * column, column < nc
    }
    else {
        const int q = pos - start[ns]; // the batched fronts' entries have an index space of their own
        bigSrc[q] = k;
        bigDst[q] = dst[k];
    }
}

// Round 5: the entries of A of the tile follow in the same launch (they used to be a launch of their own per level, k_scatter_big: 5 us on the chain of
// every level).  aPtr[d.w] .. aPtr[d.w + 1]: the tile's entries in aSrc / aDst (sorted by tile on the host, MfNumeric::setup); every entry of A lands
// in a column < nc, i.e. in a tile this kernel writes, and no two entries share a slot.
__global__ __launch_bounds__(WG) void k_extend_add(const int4* __restrict__ desc, const int* __restrict__ bigFd,
    const int* __restrict__ invMap, double* __restrict__ fronts, int ownOnly, const int* __restrict__ aPtr, const int* __restrict__ aSrc,
    const long long* __restrict__ aDst, const double* __restrict__ a)
{
    __shared__ __attribute__((aligned(16))) int fd[FD_STRIDE_EA];
    __shared__ int rmap[FUSED_MAX_KIDS_EA][TS], cmap[FUSED_MAX_KIDS_EA][TS];
    const int4 d = desc[blockIdx.x];
    const int i0 = TS * d.y, j0 = TS * d.z;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double sum[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) sum[q] = 0.0;
    int N = 0, ncOwn = 0;
    double* F = nullptr;
    for (int rec = d.x; rec >= 0;) {
        __syncthreads();
        if (tid < FD_STRIDE_EA) fd[tid] = bigFd[(size_t)rec * FD_STRIDE_EA + tid];
        __syncthreads();
        N = fd[2];
        ncOwn = fd[3];
        F = fronts + *reinterpret_cast<const long long*>(fd);
        const int nk = fd[8];
        rec = fd[9];
        for (int e = tid; e < nk * 2 * TS; e += WG) {
            const int k = e / (2 * TS), t = e - k * (2 * TS);
            const int* inv = invMap + fd[16 + 6 * k + 4];
            const int ncc = fd[16 + 6 * k + 3];
            const int I = (t < TS ? i0 : j0 - TS) + t;
            int m = -1;
            if (I < N) {
                const int ic = inv[I / 3];
                if (ic >= 0) m = ncc + 3 * ic + (I - 3 * (I / 3));
            }
            (t < TS ? rmap[k] : cmap[k] - TS)[t] = m;
        }
        __syncthreads();
        for (int k = 0; k < nk; ++k) {
            const double* __restrict__ Fc = fronts + *reinterpret_cast<const long long*>(fd + 16 + 6 * k);
            const long long Nc = fd[16 + 6 * k + 2];
            const int r = rmap[k][lane];
            double x[16];
            bool ok[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int cc = cmap[k][16 * wv + q];
                ok[q] = r >= 0 && cc >= 0 && r >= cc;
                x[q] = Fc[ok[q] ? r + Nc * cc : 0];
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) sum[q] += ok[q] ? x[q] : 0.0;
        }
    }
    const int I = i0 + lane;
    if (I < N) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int J = j0 + 16 * wv + q;
            if (J <= I && (!ownOnly || J < ncOwn)) F[I + (long long)N * J] = sum[q]; // write, not accumulate: the fronts are never zero-filled
        }
    }
    const int aBeg = aPtr[d.w], aEnd = aPtr[d.w + 1];
    if (aBeg < aEnd) { // (workgroup-uniform)
        __syncthreads(); // the tile is written: the entries of A are added by whichever thread picks them up
        for (int e = aBeg + tid; e < aEnd; e += WG) fronts[aDst[e]] += a[aSrc[e]];
    }
}

// value of `v` in lane `lane` (uniform, here always a compile-time constant after unrolling): two v_readlane_b32
__device__ __forceinline__ double bcast_lane(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// reciprocal square root to full double precision: v_rsq_f64 seed + two Newton steps (the IEEE sqrt / divide expansions are
// ~10x longer dependent chains, and this sits on the critical path of every pivot)
__device__ __forceinline__ double rsqrt_nr(double d)
{
    double r = __builtin_amdgcn_rsq(d);
    double e = fma(-d * r, r, 1.0);
    r = fma(0.5 * r, e, r);
    e = fma(-d * r, r, 1.0);
    r = fma(0.5 * r, e, r);
    return r;
}

// reciprocal to full double precision: v_rcp_f64 seed + two Newton steps
__device__ __forceinline__ double rcp_nr(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    return r;
}

// X = L11^-1 of a (<=) 32 x 32 SPD pivot block A11 = L11 L11^T by one wave, in the accumulator registers of the matrix
// cores: symmetric Gauss-Jordan, one rank-1 update per pivot as ONE v_mfma_f64_16x16x4_f64 per 16 x 16 tile.
//   blk[k * ld + r] = A(r, k), r >= k (LDS, k-major; rows / columns >= w count as identity and are not read)
//   Xs[c * LDX + r] = X(r, c), all 32 x 32 entries written (zeros above the diagonal); returns true on a pivot <= 0
// The block lives as tiles T00, T01, T11 (upper triangle: the row of a pivot is its column) in the D layout
// (row = (l >> 4) + 4 reg, col = l & 15).  Row k of a tile sits in register k >> 2 of the 16 lanes with l >> 4 == (k & 3),
// indexed by l & 15 -- which is exactly an operand of k-slice (k & 3): A[i = l & 15][kk = l >> 4], B[kk = l >> 4][j = l & 15].
// So with every other k-slice masked to zero, D[i][j] += m_i a_j needs no data movement at all: the pivot row scaled by
// -1 / d is the A operand (multipliers, rows <= k masked), the pivot row itself the B operand.  The same eliminations
// applied to W = I give the inverse of the unit-lower factor; X = D^-1/2 W.  On the dependent chain of a pivot: two
// v_readlane (d), v_rcp_f64 + two Newton steps, one multiply, one MFMA -- no square root, no lane-to-lane broadcast of
// multipliers, no LDS.  (Entries left of a pivot are dead: they only ever feed other dead entries, so neither they nor
// the B operands need masking.)  On this chip an fp64 MFMA occupies the SIMD for 64 cycles and does NOT overlap with the
// wave's own VALU work (measured: pivot time = MFMAs x 64 + ~240 cycles of chain, whatever the order) -- the matrix and
// vector fp64 rates of MI355X are the same units -- so what counts is the number of MFMAs and the length of the chain.
// Blocked 16 + 16 so that the pipe (64 cycles per fp64 MFMA here) does not become the bound:
//   pivots 0..15 update T00, T01 = U01 and W00 only (three MFMAs each; the W00 update of pivot k is issued in the middle of
//   the reciprocal chain of pivot k + 1, where the pipe would idle); then, as K = 16 products whose operands are the accumulator
//   registers as they stand (register i of a tile is row (l >> 4) + 4 i = the operand of k-step i, either side):
//       T11 -= U01^T D0^-1 U01,   W10 = -(D0^-1 U01)^T W00;
//   pivots 16..31 update T11 and W11 (two MFMAs each); finally X10 = X11 W10 (X11 transposed through LDS, where it goes anyway).
// 92 MFMAs, against 16.5 k + 3.1 k cycles for the lane-per-row Cholesky + recursive-doubling inverse this replaces.
// The same with 2 x 2 block pivots -- the version in use.  Rows k, k + 1 (k even) of a tile sit in the SAME accumulator
// register, in lane groups l >> 4 == (k & 3) and (k & 3) + 1: one MFMA applies the rank-2 update of a pivot pair (two
// k-slices instead of one), so a block costs half the MFMAs and half the dependent chains of the single-pivot sweep above
// (on this chip an fp64 MFMA occupies the SIMD for 64 cycles and does not overlap with the wave's own VALU work, so those
// two counts ARE the time).  With P = [a b; b c] the pivot block and P^-1 = [ca -cb; -cb cc], the multipliers of row i are
//     m_i^(k) = -(ca A(k, i) - cb A(k + 1, i)),    m_i^(k+1) = -(cc A(k + 1, i) - cb A(k, i)):
// "own row times own coefficient minus partner row times cb", the partner row being 16 lanes away (one ds_bpermute pair,
// issued before the reciprocal chain needs it).  The block-LDL^T factors come out as W = (unit block lower)^-1 and the pivot
// blocks; X = C^-1 W with C the 2 x 2 Cholesky factors of the pivot blocks is the (unique) inverse of the Cholesky factor:
//     X_k = W_k / l11,   X_k+1 = (W_k+1 - (l21 / l11) W_k) / l22,   l11 = sqrt(a), l21 = b / l11, l22 = sqrt(det / a).
// Blocked 16 + 16 as above: T11 -= U01^T D0^-1 U01 and W10 = -(D0^-1 U01)^T W00 with D0 the block diagonal of pivot blocks.
// the value of lane l ^ 16 (the partner row of a 2 x 2 pivot sits 16 lanes away): two ds_bpermute, in flight during the reciprocal chain.  (Round 5: gfx950's
// v_permlane16_swap_b32 does the same exchange in the VALU -- probed, correct, and no faster: 422.0 against 421.3 it/s, profiles/r05_permlane_and_border_ab.txt --
// the exchange is not what a pivot waits for.)
__device__ __forceinline__ double lane_xor16(double v) { return __shfl_xor(v, 16, 64); }

__device__ __forceinline__ bool wave_potrf_inv32_mfma(const double* blk, int ld, int w, int lane, double* Xs)
{
    const int lo = lane & 15, hi = lane >> 4;
    f64x4 T00, T01, T11, W00, W11;
    f64x4 W10 = { 0.0, 0.0, 0.0, 0.0 };
    f64x4 aV0, bV0, dV0, alV0, cbV0, aV1, bV1, dV1; // per D-layout row: pivot block entries a, b, det and the coefficients of P^-1
    const int wm = w - 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = hi + 4 * r;
        const int mn = min(q, lo), mx = max(q, lo);
        const double a00 = blk[min(mn, wm) * ld + min(mx, wm)];
        const double a01 = blk[min(q, wm) * ld + min(16 + lo, wm)];
        const double a11 = blk[min(16 + mn, wm) * ld + min(16 + mx, wm)];
        const bool dg = q == lo;
        T00[r] = (mx < w) ? a00 : (dg ? 1.0 : 0.0);
        T01[r] = (16 + lo < w) ? a01 : 0.0;
        T11[r] = (16 + mx < w) ? a11 : (dg ? 1.0 : 0.0);
        W00[r] = dg ? 1.0 : 0.0;
        W11[r] = dg ? 1.0 : 0.0;
        aV0[r] = aV1[r] = dV0[r] = dV1[r] = 1.0;
        bV0[r] = bV1[r] = alV0[r] = cbV0[r] = 0.0;
    }
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        if (k >= w) break; // wave-uniform: the rows / columns >= w are identity -- their pivots are 1, their multipliers 0, the updates add zeros (a panel
                           // of 4 columns, the last one of a 900-column separator or the second one of a 36-column leaf, takes 2 of the 16 pivot steps)
        const int h = k & 3, r = k >> 2;
        const bool selh = hi == h, both = (hi >> 1) == (h >> 1), live = both && lo > k + 1;
        const double row = T00[r];
        const double partner = lane_xor16(row);
        const double a = bcast_lane(row, 16 * h + k), b = bcast_lane(row, 16 * h + k + 1), c = bcast_lane(row, 16 * (h + 1) + k + 1);
        const double det = fma(a, c, -(b * b));
        bad |= !(a > 0.0) || !(det > 0.0); // not on the chain: a bad pivot leaves garbage behind, and the flag says so
        const double rdet = rcp_nr(det);
        const double ca = c * rdet, cb = b * rdet, cc = a * rdet;
        const double alpha = selh ? ca : cc;
        const double m = live ? fma(-alpha, row, cb * partner) : 0.0; // the B operands go in unmasked: their other k-slices meet zeros of m
        T00 = __builtin_amdgcn_mfma_f64_16x16x4f64(m, row, T00, 0, 0, 0);
        T01 = __builtin_amdgcn_mfma_f64_16x16x4f64(m, T01[r], T01, 0, 0, 0);
        W00 = __builtin_amdgcn_mfma_f64_16x16x4f64(m, W00[r], W00, 0, 0, 0);
        aV0[r] = both ? a : aV0[r];
        bV0[r] = both ? b : bV0[r];
        dV0[r] = both ? det : dV0[r];
        alV0[r] = both ? alpha : alV0[r];
        cbV0[r] = both ? cb : cbV0[r];
    }
    if (w > 16) { // wave-uniform: a block of at most 16 columns is finished (rows 16..31 are identity)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const double t = T01[ks];
            const double a = fma(-alV0[ks], t, cbV0[ks] * lane_xor16(t)); // -(D0^-1 U01)(row (l >> 4) + 4 ks, col l & 15): A operand [i = col][kk = row]
            T11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, t, T11, 0, 0, 0);
            W10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, W00[ks], W10, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            if (16 + k >= w) break;
            const int h = k & 3, r = k >> 2;
            const bool selh = hi == h, both = (hi >> 1) == (h >> 1), live = both && lo > k + 1;
            const double row = T11[r];
            const double partner = lane_xor16(row);
            const double a = bcast_lane(row, 16 * h + k), b = bcast_lane(row, 16 * h + k + 1), c = bcast_lane(row, 16 * (h + 1) + k + 1);
            const double det = fma(a, c, -(b * b));
            bad |= !(a > 0.0) || !(det > 0.0);
            const double rdet = rcp_nr(det);
            const double ca = c * rdet, cb = b * rdet, cc = a * rdet;
            const double alpha = selh ? ca : cc;
            const double m = live ? fma(-alpha, row, cb * partner) : 0.0;
            T11 = __builtin_amdgcn_mfma_f64_16x16x4f64(m, row, T11, 0, 0, 0);
            W11 = __builtin_amdgcn_mfma_f64_16x16x4f64(m, W11[r], W11, 0, 0, 0);
            aV1[r] = both ? a : aV1[r];
            bV1[r] = both ? b : bV1[r];
            dV1[r] = both ? det : dV1[r];
        }
    }
    const bool even = (hi & 1) == 0; // rows k (even) of the pairs
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = hi + 4 * r;
        const double ia0 = rsqrt_nr(aV0[r]), iv0 = ia0 * ia0, il0 = rsqrt_nr(dV0[r] * iv0);
        const double ia1 = rsqrt_nr(aV1[r]), iv1 = ia1 * ia1, il1 = rsqrt_nr(dV1[r] * iv1);
        const double p0 = even ? ia0 : il0, q0 = even ? 0.0 : -bV0[r] * iv0 * il0;
        const double p1 = even ? ia1 : il1, q1 = even ? 0.0 : -bV1[r] * iv1 * il1;
        const double w0 = W00[r], w1 = W11[r];
        Xs[lo * LDX + q] = fma(q0, lane_xor16(w0), p0 * w0);
        Xs[(16 + lo) * LDX + 16 + q] = fma(q1, lane_xor16(w1), p1 * w1);
        Xs[(16 + lo) * LDX + q] = 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // X10 = X11 W10: A[i = l & 15][kk] = X(16 + i, 16 + kk) read back transposed, B = W10 as it sits in the accumulators
    f64x4 X10 = { 0.0, 0.0, 0.0, 0.0 };
    if (w > 16) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            X10 = __builtin_amdgcn_mfma_f64_16x16x4f64(Xs[(16 + 4 * ks + hi) * LDX + 16 + lo], W10[ks], X10, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Xs[lo * LDX + 16 + hi + 4 * r] = X10[r];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return bad;
}

// Fused path of the fronts whose nc own columns fit into LDS (every front of the lower tree levels): ONE workgroup assembles
// the front (children's update matrices gathered through their inverse index maps + the entries of A), factors the nc columns
// in LDS and writes the factor panel and the Schur complement -- each exactly once.  The front never exists in HBM in its
// assembled form: no zero-fill, no extend-add pass, no read-modify-write of the update matrix.  (Before: memset + scatter +
// extend-add + factor kernels, three to four passes over every front of the lower levels, which cost more than the top of
// the tree.)
//   P[k * N + r] = column k (< nc) of the front, rows 0..N (k-major: a wave reads 64 consecutive rows)
//   cm[q * N + I] = scalar index of parent-local row I inside child q's front, or -1
constexpr int FUSED_MAX_KIDS = FUSED_MAX_KIDS_EA;
// Host-packed descriptor of a fused front, 64 ints: everything the kernel would otherwise chase through five rounds of
// dependent loads (front list -> index pointers -> child list -> child pointers -> inverse maps) arrives in one.
//   [0,1] front offset  [2] N  [3] nc  [4,5] first dinv block  [6] aBeg  [7] aEnd  [8] #children
//   child q at 16 + 6 q: [0,1] front offset  [2] N  [3] nc  [4] offset of its inverse map
constexpr int FD_STRIDE = FD_STRIDE_EA;
// Diagnosis build only (-DMF_FUSED_PROBE, tools/gpu_r5_call11.sh): thread 0 of every fused front stamps the phases of its workgroup with the 100 MHz wall clock;
// MfNumeric::factorize prints the level averages once.  Expands to nothing in the product build.
#ifndef MF_FUSED_SCHUR_GROUP
#define MF_FUSED_SCHUR_GROUP 2 // Schur tiles a wave of the fused kernel gathers together (see the end of k_front_fused; 1 tile, 1 child at a time: 423.2, 2: 425.3, 4: 422.7 it/s)
#endif
#ifdef MF_FUSED_PROBE
__device__ unsigned long long g_probe[16 << 16];
__device__ const int* g_probeBase;
#define MF_PROBE(slot) do { if (tid == 0) g_probe[16 * probeId + (slot)] = wall_clock64(); } while (0)
#define MF_PROBE_ACC(var) do { if (tid == 0) { const unsigned long long now_ = wall_clock64(); var += now_ - probeLast; probeLast = now_; } } while (0)
#else
#define MF_PROBE(slot) do {} while (0)
#define MF_PROBE_ACC(var) do {} while (0)
#endif
template <int NT>
__global__ __launch_bounds__(NT, 3) void k_front_fused(const int* __restrict__ fdesc, const int* __restrict__ invMap,
    const int* __restrict__ aLoc, const double* __restrict__ aP, double* __restrict__ fronts,
    double* __restrict__ dinv, int* __restrict__ flag)
{
    extern __shared__ double P[];
    __shared__ double Xs[NB * LDX]; // inverse of the current pivot block
    __shared__ __attribute__((aligned(16))) int fd[FD_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < FD_STRIDE) fd[tid] = fdesc[(size_t)blockIdx.x * FD_STRIDE + tid];
    __syncthreads();
    const int N = fd[2], nc = fd[3], nk = fd[8];
    double* F = fronts + *reinterpret_cast<const long long*>(fd);
    double* dblk = dinv + *reinterpret_cast<const long long*>(fd + 4) * (NB * NB);
    const int aBeg = fd[6], aEnd = fd[7];
    int* cm = reinterpret_cast<int*>(P + (size_t)nc * N);
    bool bad = false;
#ifdef MF_FUSED_PROBE
    const long long probeId = ((fdesc - g_probeBase) / FD_STRIDE + blockIdx.x) & 0xffff;
    unsigned long long probeLast = 0, probePivot = 0, probeRows = 0, probeTrail = 0, probeGather = 0, probeTiles = 0;
    if (tid == 0) {
        g_probe[16 * probeId + 12] = N;
        g_probe[16 * probeId + 13] = nc;
        g_probe[16 * probeId + 14] = nk;
    }
#endif
    MF_PROBE(0);
    // ---- index maps of the children
    for (int q = 0; q < nk; ++q) {
        const int* inv = invMap + fd[16 + 6 * q + 4];
        const int ncc = fd[16 + 6 * q + 3];
        for (int I = tid; I < N; I += NT) {
            const int In = I / 3;
            const int ic = inv[In];
            cm[q * N + I] = ic >= 0 ? ncc + 3 * ic + (I - 3 * In) : -1;
        }
    }
    __syncthreads();
    MF_PROBE(1);
    // ---- own columns: children sums (lower triangle), zeros above the diagonal.  The loads are unconditional (clamped
    // address, value selected afterwards) so that all of them are in flight together.
    // Two columns per wave and round, the children two at a time: 16 loads in flight (one column and one child per round was a
    // dependent memory round trip per child, column and strip -- the longest phase of a small front).
    for (int J0 = 2 * wv; J0 < nc; J0 += 2 * (NT / 64)) {
        const int Ja = J0, Jb = min(J0 + 1, nc - 1);
        for (int I0 = 0; I0 < N; I0 += 4 * 64) {
            double va[4] = { 0.0, 0.0, 0.0, 0.0 }, vb[4] = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll 2
            for (int q = 0; q < nk; ++q) {
                const double* Fc = fronts + *reinterpret_cast<const long long*>(fd + 16 + 6 * q);
                const long long Nc = fd[16 + 6 * q + 2];
                const int cca = cm[q * N + Ja], ccb = cm[q * N + Jb];
                double xa[4], xb[4];
                bool oka[4], okb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int I = I0 + 64 * u + lane;
                    const int r = (I < N) ? cm[q * N + I] : -1;
                    oka[u] = r >= 0 && cca >= 0 && I >= Ja;
                    okb[u] = r >= 0 && ccb >= 0 && I >= Jb;
                    xa[u] = Fc[oka[u] ? r + Nc * cca : 0];
                    xb[u] = Fc[okb[u] ? r + Nc * ccb : 0];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    va[u] += oka[u] ? xa[u] : 0.0;
                    vb[u] += okb[u] ? xb[u] : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int I = I0 + 64 * u + lane;
                if (I < N) {
                    P[Ja * N + I] = va[u];
                    if (J0 + 1 < nc) P[Jb * N + I] = vb[u];
                }
            }
        }
    }
    __syncthreads();
    MF_PROBE(2);
    // ---- entries of A (every destination is distinct)
    // aP: the values of A gathered into front order (k_gather_a); four (location, value) pairs requested per round
    for (int e0 = aBeg; e0 < aEnd; e0 += 4 * NT) {
        int loc[4];
        double av[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = min(e0 + u * NT + tid, aEnd - 1);
            loc[u] = aLoc[e];
            av[u] = aP[e];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e0 + u * NT + tid < aEnd) P[loc[u]] += av[u];
    }
    __syncthreads();
    MF_PROBE(3);
#ifdef MF_FUSED_PROBE
    probeLast = wall_clock64();
#endif

    // ---- factor the nc columns, 32 at a time, right-looking inside LDS
    for (int kb = 0; kb < nc; kb += NB, dblk += NB * NB) {
        const int w = min(NB, nc - kb);
        double* Pk = P + (size_t)kb * N + kb; // Pk[k * N + q] = F(kb + q, kb + k)
        // pivot block: X = L11^-1 straight from the Gauss-Jordan sweep in the matrix-core accumulators (see
        // wave_potrf_inv32_mfma); L11 itself is needed nowhere -- the rows below are a product with X, the solves multiply by X
        if (tid < 64) {
            __builtin_amdgcn_s_setprio(3);
            bad |= wave_potrf_inv32_mfma(Pk, N, w, tid, Xs);
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        MF_PROBE_ACC(probePivot);
        for (int e = tid; e < NB * NB; e += NT) dblk[e] = Xs[(e >> 5) * LDX + (e & 31)]; // column-major, identity-padded
        // rows below the pivot block: L21 = A21 L11^-T, formed transposed per 16-row tile on the matrix cores, in place in LDS:
        //   D(n, m) = sum_k X(n, k) A21(m, k)     A[i = l & 15][kk = l >> 4] = X(n, k) (LDS), B[kk][j] = P[(kb + k) N + row m]
        // (a lane's B entries are 16 consecutive rows of one column of P: conflict-free; X is lower triangular, so the first 16
        // result rows need k < 16 only).  One row per thread by substitution (row_trsm32_lean) was 8 us per panel.
        {
            const int lo = lane & 15, hi = lane >> 4;
            const int R0 = kb + w, ntile = (N - R0 + 15) >> 4;
            for (int t = wv; t < ntile; t += NT / 64) {
                const int row = min(R0 + 16 * t + lo, N - 1);
                double bv[8];
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int k = 4 * ks + hi;
                    const double v = P[(kb + min(k, w - 1)) * N + row];
                    bv[ks] = (k < w) ? v : 0.0;
                }
                f64x4 x0 = { 0.0, 0.0, 0.0, 0.0 }, x1 = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Xs[(4 * ks + hi) * LDX + lo], bv[ks], x0, 0, 0, 0);
                if (w > 16) {
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks)
                        x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Xs[(4 * ks + hi) * LDX + 16 + lo], bv[ks], x1, 0, 0, 0);
                }
                if (R0 + 16 * t + lo < N) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int n0 = hi + 4 * i, n1 = 16 + hi + 4 * i;
                        if (n0 < w) P[(kb + n0) * N + row] = x0[i];
                        if (n1 < w) P[(kb + n1) * N + row] = x1[i];
                    }
                }
            }
        }
        __syncthreads();
        MF_PROBE_ACC(probeRows);
        // the own columns to the right of this panel (rows >= column): 4 x 4 register tiles, operands and result in LDS
        const int c0 = kb + w;
        if (c0 < nc) {
            const int ntc = (nc - c0 + 3) >> 2, ntr = (N - c0 + 3) >> 2;
            for (int t = tid; t < ntc * ntr; t += NT) {
                const int tc = t / ntr, tr = t - tc * ntr;
                if (tr < tc) continue;
                const int i0 = c0 + 4 * tr, j0 = c0 + 4 * tc;
                double acc[4][4];
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) acc[ii][jj] = 0.0;
                for (int k = 0; k < w; ++k) {
                    const double* pk = P + (size_t)(kb + k) * N;
                    double av[4], bv[4];
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        av[ii] = pk[min(i0 + ii, N - 1)];
                        bv[ii] = pk[min(j0 + ii, N - 1)];
                    }
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) acc[ii][jj] += av[ii] * bv[jj];
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int j = j0 + jj;
                    if (j >= nc) continue;
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const int i = i0 + ii;
                        if (i < N && i >= j) P[(size_t)j * N + i] -= acc[ii][jj];
                    }
                }
            }
            __syncthreads();
        }
        MF_PROBE_ACC(probeTrail);
    }
    MF_PROBE(4);
    // ---- the factor panel goes to HBM once (the solves read it)
    for (int J = wv; J < nc; J += NT / 64)
        for (int I = J + lane; I < N; I += 64) F[I + (long long)N * J] = P[J * N + I];
    MF_PROBE(5);
    // ---- Schur complement: S = (children) - L21 L21^T, written once.  Round 5: 16 x 16 tiles on the matrix cores, one tile per wave at a time, operands
    // straight from the panel in LDS (a lane's A / B entry: 16 consecutive rows of one column of P -- conflict-free unless N is a multiple of 32).  As 4 x 4
    // register tiles per thread (rounds 1-4) this block was bound by its LDS reads -- eight ds_read_b64 per sixteen multiply-adds: 18 us for a front of
    // 250 rows and 60 columns, most of what a workgroup of the levels just below the batched ones took.  v_mfma_f64_16x16x4_f64: A[l & 15][l >> 4] =
    // L(j0 + (l & 15), k), B[l >> 4][l & 15] = L(i0 + (l & 15), k); D register r of lane l = S(i0 + (l & 15), j0 + (l >> 4) + 4 r): the 16 lanes of an
    // accumulator row store 16 consecutive rows of one column, 128 contiguous bytes.
    // Tiles go through a wave SG at a time: the gathers of SG tiles and of two children are requested together, then summed child by child in the order they always
    // were (bit-identical sums), then the SG products, then the stores.  Round 5, after the phase probe of the kernel (-DMF_FUSED_PROBE, profiles/r05_fused_front_phases.txt):
    // the Schur block is 16-27 us of the 29-43 us a front of levels 1-3 takes, and 9-12 us of that are these gathers WHATEVER their grouping -- four times fewer dependent
    // rounds took four times as long each: the phase is bound by what one CU can pull through its L1 (three fronts share it: ~27 KB of children per front, touched as
    // partially used 128-byte lines), not by the latency of a round.  Groups of two are the measured optimum (fewer registers than four, fewer rounds than one).
    {
        const int mt = N - nc;
        const int nt16 = (mt + 15) >> 4;
        const int nTiles = nt16 * (nt16 + 1) / 2;
        const int lo = lane & 15, hi = lane >> 4;
        const int ksteps = (nc + 3) >> 2;
        constexpr int SG = MF_FUSED_SCHUR_GROUP, NW = NT / 64;
        for (int tb = wv; tb < nTiles; tb += NW * SG) {
            int i0[SG], j0[SG];
            bool live[SG];
#pragma unroll
            for (int u = 0; u < SG; ++u) {
                const int t = tb + u * NW;
                live[u] = t < nTiles;
                const int tt = live[u] ? t : tb;
                int tr = (int)((sqrtf(8.0f * (float)tt + 1.0f) - 1.0f) * 0.5f); // tile (tr, tc), tc <= tr, number tt of the lower triangle row by row
                while (tr * (tr + 1) / 2 > tt) --tr;
                while ((tr + 1) * (tr + 2) / 2 <= tt) ++tr;
                const int tc = tt - tr * (tr + 1) / 2;
                i0[u] = nc + 16 * tr;
                j0[u] = nc + 16 * tc;
            }
#ifdef MF_FUSED_PROBE
            probeLast = wall_clock64();
#endif
            double ch[SG][4];
#pragma unroll
            for (int u = 0; u < SG; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) ch[u][r] = 0.0;
            for (int q = 0; q < nk; q += 2) {
                const bool two = q + 1 < nk;
                const int qb = two ? q + 1 : q;
                const double* Fa = fronts + *reinterpret_cast<const long long*>(fd + 16 + 6 * q);
                const double* Fb = fronts + *reinterpret_cast<const long long*>(fd + 16 + 6 * qb);
                const long long Na = fd[16 + 6 * q + 2], Nb = fd[16 + 6 * qb + 2];
                const int* ma = cm + q * N;
                const int* mb = cm + qb * N;
                double xa[SG][4], xb[SG][4];
                bool oka[SG][4], okb[SG][4];
#pragma unroll
                for (int u = 0; u < SG; ++u) {
                    const int row = i0[u] + lo;
                    const int ra = (row < N) ? ma[row] : -1, rb = (row < N) ? mb[row] : -1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = j0[u] + hi + 4 * r;
                        const int ca = (j < N) ? ma[j] : -1, cb = (j < N) ? mb[j] : -1;
                        oka[u][r] = ra >= 0 && ca >= 0 && ra >= ca;
                        okb[u][r] = two && rb >= 0 && cb >= 0 && rb >= cb;
                        xa[u][r] = Fa[oka[u][r] ? ra + Na * ca : 0];
                        xb[u][r] = Fb[okb[u][r] ? rb + Nb * cb : 0];
                    }
                }
#pragma unroll
                for (int u = 0; u < SG; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ch[u][r] += oka[u][r] ? xa[u][r] : 0.0;
                        ch[u][r] += okb[u][r] ? xb[u][r] : 0.0;
                    }
            }
#ifdef MF_FUSED_PROBE
            if (tid == 0) { // the gathers are waited for here (in the product build they would be waited for inside the sums above)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long now_ = wall_clock64();
                probeGather += now_ - probeLast;
                probeLast = now_;
            }
#endif
#pragma unroll
            for (int u = 0; u < SG; ++u) {
                if (!live[u]) continue; // wave-uniform
                const int row = i0[u] + lo;
                f64x4 acc = { 0.0, 0.0, 0.0, 0.0 };
                const double* pa = P + min(j0[u] + lo, N - 1); // rows past the end of the front are clamped: their products land in entries that are never written
                const double* pb = P + min(row, N - 1);
                for (int ks0 = 0; ks0 < ksteps; ks0 += 4) { // four k-steps at a time: their eight LDS reads are in flight before the first product
                    double av[4], bv[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int k = 4 * (ks0 + v) + hi;
                        const size_t off = (size_t)min(k, nc - 1) * N;
                        av[v] = pa[off];
                        bv[v] = pb[off];
                    }
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc = __builtin_amdgcn_mfma_f64_16x16x4f64((4 * (ks0 + v) + hi < nc) ? av[v] : 0.0, bv[v], acc, 0, 0, 0);
                }
                if (row < N) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = j0[u] + hi + 4 * r;
                        if (j < N && row >= j) F[row + (long long)N * j] = ch[u][r] - acc[r];
                    }
                }
            }
            MF_PROBE_ACC(probeTiles);
        }
    }
    if (bad) atomicOr(flag, 1);
#ifdef MF_FUSED_PROBE
    __syncthreads();
    MF_PROBE(6);
    if (tid == 0) {
        g_probe[16 * probeId + 8] = probePivot;
        g_probe[16 * probeId + 9] = probeRows;
        g_probe[16 * probeId + 10] = probeTrail;
        g_probe[16 * probeId + 11] = probeGather;
        g_probe[16 * probeId + 15] = probeTiles;
    }
#endif
}

// Schur complement of a big front in one pass: S(i, j) -= sum_{c < nc} L(i, c) L(j, c) for i, j >= nc.  Doing this per
// 32-column step (right-looking) re-reads and re-writes the whole update matrix every step, which made the middle levels
// of the tree HBM-bound; here every tile is read-modify-written once.  desc = (front, ti, tj, 0), 32 x 32 tiles, ti >= tj.
// One workgroup per 32 x 32 tile of S, four waves that split the nc columns of L between them (chunk c goes to wave c mod 4)
// and combine through LDS in a fixed order at the end.  Upper levels of the tree have a handful of fronts: with 64 x 64
// tiles and a serial loop over all of nc, a level was a few dozen workgroups each waiting out nc / 32 dependent
// load -> multiply rounds.  The MFMA operands come straight from the front (a lane's A / B entry is one double of L; 16
// lanes read 16 consecutive rows), so the loop has no LDS staging and no barrier.  The product is formed transposed,
// D(j, i): the 16 lanes of an accumulator row hold 16 consecutive rows i of one column j, which makes the read-modify-write
// of the column-major front 128-byte contiguous.  v_mfma_f64_16x16x4_f64: A[l & 15][l >> 4], B[l >> 4][l & 15],
// D column = l & 15, row = (l >> 4) + 4 reg.
constexpr int TQ = 32; // Schur tile
// [cLo, cHi): the columns of L of this pass (multiples of 32).  One pass over all columns behind the chain (k_big_schur) on most levels; on the top
// levels, where the pivot chain is what the level takes, the update rides on the chain's own launches in passes of a few panels (role S of k_big_step,
// round 4): only the last pass is left behind the chain.  red: 4096 doubles of LDS, [wave][16 x 16 tile][D layout: 64 lanes x 4].
__device__ __forceinline__ void schur_tile32(const int N, const int ncAll, double* __restrict__ F, const int ti, const int tj, const int cLo, const int cHi,
    double (*red)[4][256])
{
    const int nc = min(ncAll, cHi);
    if (cLo >= nc) return; // a front with fewer panels than the widest of its level: nothing left for this pass
    const int tid = threadIdx.x;
    const int i0 = ncAll + TQ * ti, j0 = ncAll + TQ * tj; // the tile sits behind ALL own columns of the front; nc is the end of this pass
    const int wv = tid >> 6, l = tid & 63;
    const int ar = l & 15, ak = l >> 4;
    // the entries this wave will update at the end (16 x 16 tile wv of the 32 x 32 tile) are requested now: as a
    // read-modify-write at the end they were four dependent memory round trips behind the reduction
    double old[4];
    {
        const int row = min(i0 + 16 * (wv & 1) + ar, N - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) old[r] = F[row + (long long)N * min(j0 + 16 * (wv >> 1) + ak + 4 * r, N - 1)];
    }
    f64x4 acc[2][2]; // [nj][mi]: rows j of D, columns i
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{ 0.0, 0.0, 0.0, 0.0 };
    // rows past the end of the front are clamped: their products land in entries that are never written
    const double* pa0 = F + min(i0 + ar, N - 1);
    const double* pa1 = F + min(i0 + 16 + ar, N - 1);
    const double* pb0 = F + min(j0 + ar, N - 1);
    const double* pb1 = F + min(j0 + 16 + ar, N - 1);
    // The nc columns go to the four waves in chunks of CW.  The loads of a wave's next chunk are issued before the products
    // of the current one and only touched after them (the compiler neither pipelines the loop nor keeps that order by itself:
    // hence the scheduling fences).  Loads are unconditional with a clamped column; past nc only the A operands need zeroing.
    constexpr int CW = 16, CS = CW / 4;
    const int nch = (nc + CW - 1) / CW, ch0 = cLo / CW;
    double a0[CS], a1[CS], b0[CS], b1[CS], na0[CS], na1[CS], nb0[CS], nb1[CS];
    auto fetch = [&](int ch, double* x0, double* x1, double* y0, double* y1) {
#pragma unroll
        for (int ks = 0; ks < CS; ++ks) {
            const long long off = (long long)N * min(CW * ch + 4 * ks + ak, nc - 1);
            x0[ks] = pa0[off];
            x1[ks] = pa1[off];
            y0[ks] = pb0[off];
            y1[ks] = pb1[off];
        }
    };
    auto mult = [&](int ch, const double* x0, const double* x1, const double* y0, const double* y1) {
#pragma unroll
        for (int ks = 0; ks < CS; ++ks) {
            const bool in = CW * ch + 4 * ks + ak < nc;
            const double m0 = in ? y0[ks] : 0.0, m1 = in ? y1[ks] : 0.0;
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(m0, x0[ks], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(m0, x1[ks], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(m1, x0[ks], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(m1, x1[ks], acc[1][1], 0, 0, 0);
        }
    };
    // two register sets, ping-pong (a copy from a "next" set would wait for its loads at the end of every iteration)
    // the fetches are unconditional (past the end they re-read the clamped last column): behind a branch, the compiler's wait
    // counters have to assume the loads were not issued and the products end up waiting for the newest load
    fetch(ch0 + wv, a0, a1, b0, b1);
    for (int ch = ch0 + wv; ch < nch; ch += 8) {
        fetch(ch + 4, na0, na1, nb0, nb1);
        __builtin_amdgcn_sched_barrier(0);
        mult(ch, a0, a1, b0, b1);
        __builtin_amdgcn_sched_barrier(0);
        fetch(ch + 8, a0, a1, b0, b1);
        __builtin_amdgcn_sched_barrier(0);
        if (ch + 4 < nch) mult(ch + 4, na0, na1, nb0, nb1);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wv][2 * nj + mi][64 * r + l] = acc[nj][mi][r];
    __syncthreads();
    // wave q finishes 16 x 16 tile q = 2 nj + mi
    const int nj = wv >> 1, mi = wv & 1;
    const int row = i0 + 16 * mi + ar;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int col = j0 + 16 * nj + ak + 4 * r;
        const double v = ((red[0][wv][64 * r + l] + red[1][wv][64 * r + l]) + red[2][wv][64 * r + l]) + red[3][wv][64 * r + l];
        if (col < N && row < N && row >= col) F[row + (long long)N * col] = old[r] - v;
    }
}
__global__ __launch_bounds__(WG, MF_SCHUR_OCC) void k_big_schur(const int4* __restrict__ desc, double* __restrict__ fronts)
{
    __shared__ double red[4][4][256];
    // two records per workgroup (see k_big_step): (front, ti, tj, 0) and (N, nc, front offset)
    const int4 d = desc[2 * blockIdx.x];
    const int4 d2 = desc[2 * blockIdx.x + 1];
    schur_tile32(d2.x, d2.y, fronts + (((long long)(unsigned)d2.w << 32) | (unsigned)d2.z), d.y, d.z, 0, 1 << 30, red);
}

// The same update with 64 x 64 tiles, for levels of the tree that have thousands of tiles anyway: one wave per 32 x 32 quadrant running over
// ALL nc columns, no split of the columns over the waves and so no reduction through LDS, no barrier; half the operand loads per flop of the
// 32 x 32 version (a workgroup reads 128 rows of L for 4096 entries of S instead of 64 for 1024).  The fronts of those levels have nc of
// 100-450: split four ways a wave ran two or three chunks between its prologue and the LDS reduction.  Upper levels keep the 32 x 32 kernel: they
// have a handful of fronts and need the tiles for parallelism.  desc as above with 64 x 64 tile indices.
constexpr int TQ64 = 64;
// LOAD_OLD = false: `old` arrives filled (the children's sums gathered by the caller: k_big_schur64_ea) and the tile is WRITTEN, not read-modify-written.
// (Round 5: parking `old` in LDS over the product loop -- 32 registers less, a third wave per SIMD -- changed nothing, measured at 45 K and 375 K nodes:
// the kernel is not occupancy-bound.  profiles/r05_solver_ab_xcd_occupancy.txt.)
// orig: first row / column of tile (0, 0); colEnd: columns from here on are not this pass's (the Schur passes: orig = nc, colEnd = N; the bulk update of the wide
// fronts' own columns, k_big_bulk: orig = the end of the outer block, colEnd = nc); [cLo, nc): the columns of L of this pass.
template <bool LOAD_OLD>
__device__ __forceinline__ void schur_tile64_core(const int N, const int orig, const int colEnd, const int nc, double* __restrict__ F, const int ti, const int tj,
    const int cLo, double (&old)[2][2][4])
{
    if (cLo >= nc) return;
    const int tid = threadIdx.x;
    const int wv = tid >> 6, l = tid & 63;
    const int i0 = orig + TQ64 * ti + 32 * (wv & 1), j0 = orig + TQ64 * tj + 32 * (wv >> 1); // this wave's quadrant
    if (i0 + 31 < j0 || i0 >= N || j0 >= colEnd) return; // entirely above the diagonal (the upper right quadrant of a diagonal tile) or outside
    const int ar = l & 15, ak = l >> 4;
    if (LOAD_OLD) {
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int row = min(i0 + 16 * mi + ar, N - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) old[nj][mi][r] = F[row + (long long)N * min(j0 + 16 * nj + ak + 4 * r, N - 1)];
            }
    }
    f64x4 acc[2][2]; // [nj][mi]: rows j of D, columns i (formed transposed, see k_big_schur)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{ 0.0, 0.0, 0.0, 0.0 };
    const double* pa0 = F + min(i0 + ar, N - 1);
    const double* pa1 = F + min(i0 + 16 + ar, N - 1);
    const double* pb0 = F + min(j0 + ar, N - 1);
    const double* pb1 = F + min(j0 + 16 + ar, N - 1);
    constexpr int CW = 16, CS = CW / 4;
    const int nch = (nc + CW - 1) / CW;
    double a0[CS], a1[CS], b0[CS], b1[CS], na0[CS], na1[CS], nb0[CS], nb1[CS];
    auto fetch = [&](int ch, double* x0, double* x1, double* y0, double* y1) {
#pragma unroll
        for (int ks = 0; ks < CS; ++ks) {
            const long long off = (long long)N * min(CW * ch + 4 * ks + ak, nc - 1);
            x0[ks] = pa0[off];
            x1[ks] = pa1[off];
            y0[ks] = pb0[off];
            y1[ks] = pb1[off];
        }
    };
    auto mult = [&](int ch, const double* x0, const double* x1, const double* y0, const double* y1) {
#pragma unroll
        for (int ks = 0; ks < CS; ++ks) {
            const bool in = CW * ch + 4 * ks + ak < nc;
            const double m0 = in ? y0[ks] : 0.0, m1 = in ? y1[ks] : 0.0;
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(m0, x0[ks], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(m0, x1[ks], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(m1, x0[ks], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(m1, x1[ks], acc[1][1], 0, 0, 0);
        }
    };
    // ping-pong register sets, unconditional fetches behind scheduling fences: see k_big_schur
    const int ch0 = cLo / CW;
    fetch(ch0, a0, a1, b0, b1);
    for (int ch = ch0; ch < nch; ch += 2) {
        fetch(ch + 1, na0, na1, nb0, nb1);
        __builtin_amdgcn_sched_barrier(0);
        mult(ch, a0, a1, b0, b1);
        __builtin_amdgcn_sched_barrier(0);
        fetch(ch + 2, a0, a1, b0, b1);
        __builtin_amdgcn_sched_barrier(0);
        if (ch + 1 < nch) mult(ch + 1, na0, na1, nb0, nb1);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int row = i0 + 16 * mi + ar;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = j0 + 16 * nj + ak + 4 * r;
                if (col < colEnd && row < N && row >= col) F[row + (long long)N * col] = old[nj][mi][r] - acc[nj][mi][r];
            }
        }
}
__device__ __forceinline__ void schur_tile64(const int N, const int ncAll, double* __restrict__ F, const int ti, const int tj, const int cLo, const int cHi)
{
    double old[2][2][4];
    schur_tile64_core<true>(N, ncAll, N, min(ncAll, cHi), F, ti, tj, cLo, old);
}
__global__ __launch_bounds__(WG, MF_SCHUR_OCC) void k_big_schur64(const int4* __restrict__ desc, double* __restrict__ fronts)
{
    const int4 d = desc[2 * blockIdx.x];
    const int4 d2 = desc[2 * blockIdx.x + 1];
    schur_tile64(d2.x, d2.y, fronts + (((long long)(unsigned)d2.w << 32) | (unsigned)d2.z), d.y, d.z, 0, 1 << 30);
}

// Two-level blocking of the WIDE fronts (round 5; nc >= bulkMinNc_, the top separators of meshes beyond ~200 K nodes).  A step launch applies its panel as a rank-32
// update -- 2.7 flops per byte of the trailing matrix it reads and writes -- and did so to ALL own columns behind the panel: at 375 K nodes the step launches ran at 12 %
// of the fp64 peak and were half of the factorisation.  Here the own columns go in outer blocks of OBW: a step's rank-32 update stops at the end of its outer block
// (role A records carry that end in place of nc), and when the block's last panel is final ONE launch of this kernel applies all OBW columns of the block to the own
// columns behind it, rows down to N -- the Schur kernel's 64 x 64 tile product over a column range, eight times the flops per byte.  The panel that opens the next block
// then has nothing left to apply (role B, kb <= -2).  desc = (OBW, cLo, ti, tj) + (N, nc, front offset): tile (ti, tj) counted from E = min(nc, cLo + OBW).
// Which fronts: those of a level whose step launches move enough bytes for this to pay (setup(): bulkMinMB_), not the lone root of a small mesh -- there the
// extra launches on the chain cost more than the traffic they save (45 K nodes: 1.84 -> 1.88 ms with the root's 900 columns in blocks).
__global__ __launch_bounds__(WG, MF_SCHUR_OCC) void k_big_bulk(const int4* __restrict__ desc, double* __restrict__ fronts)
{
    const int4 d = desc[2 * blockIdx.x];
    const int4 d2 = desc[2 * blockIdx.x + 1];
    const int N = d2.x, nc = d2.y, E = min(nc, d.y + d.x);
    double old[2][2][4];
    schur_tile64_core<true>(N, E, nc, E, fronts + (((long long)(unsigned)d2.w << 32) | (unsigned)d2.z), d.z, d.w, d.y, old);
}

// The same with the EXTEND-ADD of the update block fused in (round 4): S = (children) - L21 L21^T written once.  The extend-add kernel of such a level
// only writes the front's own columns; the update block used to be written by it (children sums or zeros), read back and written again here: two passes
// over the largest part of every middle-level front.  desc = (front, ti, tj, packed record of the front's children) + (N, nc, front offset); the
// records, the child -> parent index maps and the order of the sums are those of k_extend_add.
__global__ __launch_bounds__(WG) void k_big_schur64_ea(const int4* __restrict__ desc, const int* __restrict__ bigFd, const int* __restrict__ invMap,
    double* __restrict__ fronts)
{
    __shared__ __attribute__((aligned(16))) int fd[FD_STRIDE_EA];
    __shared__ int rmap[FUSED_MAX_KIDS_EA][TQ64], cmap[FUSED_MAX_KIDS_EA][TQ64];
    const int4 d = desc[2 * blockIdx.x];
    const int4 d2 = desc[2 * blockIdx.x + 1];
    const int N = d2.x, nc = d2.y;
    double* F = fronts + (((long long)(unsigned)d2.w << 32) | (unsigned)d2.z);
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, ar = l & 15, ak = l >> 4;
    const int I0 = nc + TQ64 * d.y, J0 = nc + TQ64 * d.z;
    const int qi = 32 * (wv & 1), qj = 32 * (wv >> 1);
    const bool active = !(I0 + qi + 31 < J0 + qj || I0 + qi >= N || J0 + qj >= N);
    double old[2][2][4];
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) old[nj][mi][r] = 0.0;
    for (int rec = d.w; rec >= 0;) {
        __syncthreads();
        if (tid < FD_STRIDE_EA) fd[tid] = bigFd[(size_t)rec * FD_STRIDE_EA + tid];
        __syncthreads();
        const int nk = fd[8];
        rec = fd[9];
        for (int e = tid; e < nk * 2 * TQ64; e += WG) {
            const int k = e / (2 * TQ64), t = e - k * (2 * TQ64);
            const int* inv = invMap + fd[16 + 6 * k + 4];
            const int ncc = fd[16 + 6 * k + 3];
            const int I = (t < TQ64 ? I0 : J0 - TQ64) + t;
            int m = -1;
            if (I < N) {
                const int ic = inv[I / 3];
                if (ic >= 0) m = ncc + 3 * ic + (I - 3 * (I / 3));
            }
            (t < TQ64 ? rmap[k] : cmap[k] - TQ64)[t] = m;
        }
        __syncthreads();
        if (active) {
            for (int k = 0; k < nk; ++k) {
                const double* __restrict__ Fc = fronts + *reinterpret_cast<const long long*>(fd + 16 + 6 * k);
                const long long Nc = fd[16 + 6 * k + 2];
                double x[2][2][4];
                bool ok[2][2][4];
#pragma unroll
                for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) {
                        const int rr = rmap[k][qi + 16 * mi + ar];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int cc = cmap[k][qj + 16 * nj + ak + 4 * r];
                            ok[nj][mi][r] = rr >= 0 && cc >= 0 && rr >= cc;
                            x[nj][mi][r] = Fc[ok[nj][mi][r] ? rr + Nc * cc : 0];
                        }
                    }
#pragma unroll
                for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int r = 0; r < 4; ++r) old[nj][mi][r] += ok[nj][mi][r] ? x[nj][mi][r] : 0.0;
            }
        }
    }
    if (active) schur_tile64_core<false>(N, nc, N, nc, F, d.y, d.z, 0, old);
}


// Role C of the step kernel (round 4): X = L11^-1 grows BY BORDERING, one panel per step launch, beside the pivot chain that produces the panels.
// Launch p + 1 holds, for every front with an explicit inverse, the rows R = [kb, kb + 32) of panel p (finished by launch p), one workgroup per
// column tile C = [c0, c0 + CT) left of them:
//     X(R, C) = -X_RR ( L(R, C..kb) X(C..kb, C) )            X(0:kb, 0:kb) was finished by the launches before, X_RR is the panel's dinv block
// Both products are formed in ONE workgroup (the column tiles are independent), so the inverse is complete one launch after the last panel
// instead of eleven dependent launches (recursive doubling on a side stream: the tail of every factorisation, 0.105 ms at mat150).
//   stage 1: T(r, c) = sum_k L(kb + r, k) X(k, c0 + c) over k in [c0, kb): four waves split k in batches of 16, combine through LDS in a fixed order
//   stage 2: X(kb + r, c0 + c) = -sum_k' Xd(r, k') T(k', c): one 16 x 16 quadrant per wave, T from LDS
// A workgroup is as long as its k range (up to nc columns, on ONE CU): tiles far left of the panel are CT = 16 columns wide (half the products per
// wave, twice the workgroups), and the operands are fetched two batches ahead of their products (one batch = 8 or 16 MFMAs = 0.2 - 0.4 us, less than
// a strided L2 round trip).  d = (front, kb, c0, -3 [CT = 32] / -6 [CT = 16]); c0 == kb: the diagonal block itself (a copy of the dinv block).
template <int CT, bool XT_ON>
__device__ __forceinline__ void step_border(const int4 d, const int4 d2, const TreeView& tv, const XinvView& xv, const double* __restrict__ F,
    const double* __restrict__ dinv, double* sm)
{
    constexpr int NA = CT / 16; // 16-column halves of the tile
    const int s = d.x, kb = d.y, c0 = d.z;
    const int N = d2.x, nc = d2.y;
    const int w = min(NB, nc - kb);
    double* X = xv.X + xv.xOff[s];
    // X^T beside X (in the scratch buffer the recursive doubling would use): the first product reads X(k, c) for 16 columns c and 4 rows k per instruction --
    // 16 cache lines from the column-major X, 4 from its transpose
    double* XT = XT_ON ? xv.T + xv.xOff[s] : nullptr;
    const double* blk = dinv + (tv.dinvOff[s] + kb / NB) * (NB * NB); // blk[c * 32 + r] = Xd(r, c), identity-padded
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, lo = l & 15, hi = l >> 4;
    if (c0 == kb) {
        for (int e = tid; e < NB * NB; e += WGB) {
            const int c = e >> 5, r = e & 31;
            if (r < w && c < w) {
                X[(kb + r) + (long long)nc * (kb + c)] = blk[e];
                if (XT_ON) XT[(kb + c) + (long long)nc * (kb + r)] = blk[e];
            }
        }
        return;
    }
    double(*red)[4][256] = reinterpret_cast<double(*)[4][256]>(sm); // [wave][16 x 16 tile][D layout: 64 lanes x 4]
    f64x4 acc[NA][2];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{ 0.0, 0.0, 0.0, 0.0 };
    const int cc[2] = { c0 + lo, c0 + 16 + lo }; // < kb: always inside
    const int rr[2] = { kb + lo, kb + 16 + lo };
    const int rrc[2] = { min(rr[0], nc - 1), min(rr[1], nc - 1) };
    // the second stage's operand (the panel's own inverse block, lower triangular) is requested now
    double xd[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) xd[ks] = blk[(4 * ks + hi) * NB + 16 * (wv & 1) + lo]; // Xd(r = 16 b + lo, k' = 4 ks + hi)
    // three operand sets in rotation, unconditional clamped fetches, masks at use: see k_big_schur / k_xinv_gemm
    auto fetch = [&](int k0, double(&ra)[NA][4], double(&rb)[2][4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kc = min(k0 + 4 * ks + hi, nc - 1);
#pragma unroll
            for (int q = 0; q < NA; ++q) ra[q][ks] = XT_ON ? XT[cc[q] + (long long)nc * kc] : X[kc + (long long)nc * cc[q]];
#pragma unroll
            for (int q = 0; q < 2; ++q) rb[q][ks] = F[rrc[q] + (long long)N * kc];
        }
    };
    auto mult = [&](int k0, const double(&ra)[NA][4], const double(&rb)[2][4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = k0 + 4 * ks + hi;
            const bool kin = k < kb;
            double ma[NA], mb[2];
#pragma unroll
            for (int q = 0; q < NA; ++q) ma[q] = (kin && k >= cc[q]) ? ra[q][ks] : 0.0;
#pragma unroll
            for (int q = 0; q < 2; ++q) mb[q] = (kin && rr[q] < kb + w) ? rb[q][ks] : 0.0;
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ma[a], mb[b], acc[a][b], 0, 0, 0);
        }
    };
    double xa[NA][4], la[2][4], xb[NA][4], lb[2][4], xc[NA][4], lc[2][4];
    int k0 = c0 + 16 * wv;
    fetch(k0, xa, la);
    fetch(k0 + 64, xb, lb);
    for (; k0 < kb; k0 += 192) {
        fetch(k0 + 128, xc, lc);
        __builtin_amdgcn_sched_barrier(0);
        mult(k0, xa, la);
        __builtin_amdgcn_sched_barrier(0);
        fetch(k0 + 192, xa, la);
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + 64 < kb) mult(k0 + 64, xb, lb);
        __builtin_amdgcn_sched_barrier(0);
        fetch(k0 + 256, xb, lb);
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + 128 < kb) mult(k0 + 128, xc, lc);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wv][2 * a + b][64 * i + l] = acc[a][b][i];
    __syncthreads();
    // wave q finishes the 16 x 16 tile q = 2 a + b of T: entry i of a lane is T(r = 16 b + lo, c = 16 a + hi + 4 i)
    const int a = wv >> 1, b = wv & 1;
    const bool mine = a < NA; // a 16-column tile has two quadrants: waves 2 and 3 only help with the sums over k
    double t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = ((red[0][wv & (2 * NA - 1)][64 * i + l] + red[1][wv & (2 * NA - 1)][64 * i + l]) + red[2][wv & (2 * NA - 1)][64 * i + l]) + red[3][wv & (2 * NA - 1)][64 * i + l];
    __syncthreads();
    double* Ts = sm; // Ts[k' * LDP + c] = T(k', c)
    if (mine) {
#pragma unroll
        for (int i = 0; i < 4; ++i) Ts[(16 * b + lo) * LDP + 16 * a + hi + 4 * i] = t[i];
    }
    __syncthreads();
    if (!mine) return;
    // stage 2, formed transposed like stage 1: D(c, r) = sum_k' T(k', c) Xd(r, k'), k' <= r
    f64x4 o = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const int kp = 4 * ks + hi;
        o = __builtin_amdgcn_mfma_f64_16x16x4f64(Ts[kp * LDP + 16 * a + lo], (kp <= 16 * b + lo) ? xd[ks] : 0.0, o, 0, 0, 0);
    }
    const int r = 16 * b + lo;
    if (r < w) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + 16 * a + hi + 4 * i;
            X[(kb + r) + (long long)nc * c] = -o[i];
            if (XT_ON) XT[c + (long long)nc * (kb + r)] = -o[i];
        }
    }
}

// ---- big fronts: level-batched 32-column steps, one launch per step ------------------------------------
// desc = (first dinv block of the front, kb of the panel being applied or -1, a, b) + (N, nc, front offset); 256 threads: three row waves + one pivot wave
//   b >= 0 : role A, trailing tile (ti, tj) = (a, b) of the matrix behind panel kb and panel kb+32
//   b == -2: role B, rows [kb1 + a, kb1 + a + 192) of the next panel (kb1 = kb + 32, or 0 when kb == -1)
// TOP = true (the levels of the top separators, where the chain of these launches IS the level) adds a role that rides on the chain's launches
// instead of following it as launches of its own:
//   b == -6: role C, one 32 x 16 tile of the explicit inverse X = L11^-1 growing by bordering (step_border): d = (front, panel, first column, b)
// (a separate instance: the extra role would cost the trailing tiles of the middle levels registers they do not need.  Tried in round 4 and removed
// in round 5, numbers in profiles/: the Schur complement folded into these launches in passes -- role S, neutral at 45 K nodes, a loss at 375 K --,
// several steps merged into one launch with in-launch counters -- 2-5 x slower: an agent-scope release / acquire is an L2 write-back / invalidate
// and was paid per workgroup, profiles/r04_merged_step_launches_ab.txt -- and two panels per launch -- k_big_step2, brought up in round 5: correct and
// 8 % slower, profiles/r05_step2_bringup.txt.)
template <bool TOP>
__device__ __forceinline__ void step_work(const int wg, const int4 d, const int4 d2, const TreeView& tv, double* __restrict__ fronts,
    double* __restrict__ dinv, int* __restrict__ flag, const XinvView& xv)
{
    __shared__ double sm[2 * NB * TS];
    // two records per workgroup, both addressed by the workgroup index alone: (first dinv block, kb, a, b) and (N, nc, front offset) -- the front's
    // dimensions used to be two more dependent loads (tree arrays indexed by the front) at the head of every step
    const int N = d2.x, nc = d2.y;
    double* F = fronts + (((long long)(unsigned)d2.w << 32) | (unsigned)d2.z);
    const int tid = threadIdx.x;
    if (TOP && d.w == -6) {
        step_border<16, true>(d, d2, tv, xv, F, dinv, sm);
        return;
    }
    const int kb = d.y; // >= 0: the panel to apply; -1: none, the next panel is the first; <= -2: none (the bulk update of its outer block did it), the next panel starts at -kb - 2
    const int w = (kb >= 0) ? min(NB, nc - kb) : 0;
    const int kb1 = (kb >= 0) ? kb + w : (kb == -1 ? 0 : -kb - 2);
    const int w1 = (kb1 < nc) ? min(NB, nc - kb1) : 0;
    if (d.w >= 0) {
        // ---- role A: F[i0.., j0..] -= P_kb[i0..] P_kb[j0..]^T behind the next panel, own columns (< nc) only.
        // Round 5: on the matrix cores, one wave per 32 x 32 quadrant of the 64 x 64 tile, operands straight from the front -- the Schur kernel's inner
        // product (schur_tile64_core) over the 32 columns of one panel.  As 4 x 4 register tiles fed from LDS (rounds 1-4) the update was bound by the LDS
        // reads: eight ds_read_b64 per sixteen multiply-adds; the two forms have the same arithmetic peak on this chip, so the win is the operand traffic.  At
        // 375 K nodes, where these tiles are most of what a step launch does, the step kernels ran at 12 % of the fp64 peak against 45 % for the Schur kernel
        // (profiles/r05_mat433_kernel_stats.md).
        const int M0 = kb1 + w1;
        const int wv = tid >> 6, l = tid & 63, ar = l & 15, ak = l >> 4;
        const int ia = M0 + TS * d.z + 32 * (wv & 1), ja = M0 + TS * d.w + 32 * (wv >> 1); // this wave's quadrant
        const int colEnd = min(N, nc); // columns >= nc form the Schur complement: one pass at the end (k_big_schur*)
        if (ia + 31 < ja || ia >= N || ja >= colEnd) return; // entirely above the diagonal, below the front or behind the own columns (no barrier follows)
        const double* pa0 = F + min(ia + ar, N - 1);
        const double* pa1 = F + min(ia + 16 + ar, N - 1);
        const double* pb0 = F + min(ja + ar, N - 1);
        const double* pb1 = F + min(ja + 16 + ar, N - 1);
        // all 32 operand loads and the 16 old values of the quadrant are issued before anything waits on them
        double x0[NB / 4], x1[NB / 4], y0[NB / 4], y1[NB / 4], old[2][2][4];
#pragma unroll
        for (int ks = 0; ks < NB / 4; ++ks) {
            const long long off = (long long)N * (kb + min(4 * ks + ak, w - 1)); // role A only exists behind a panel: w >= 1
            x0[ks] = pa0[off];
            x1[ks] = pa1[off];
            y0[ks] = pb0[off];
            y1[ks] = pb1[off];
        }
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int row = min(ia + 16 * mi + ar, N - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) old[nj][mi][r] = F[row + (long long)N * min(ja + 16 * nj + ak + 4 * r, N - 1)];
            }
        __builtin_amdgcn_sched_barrier(0);
        f64x4 acc[2][2]; // [nj][mi]: rows j of D, columns i (formed transposed: 16 lanes of an accumulator row hold 16 consecutive rows i of one column j)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{ 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
        for (int ks = 0; ks < NB / 4; ++ks) {
            const bool in = 4 * ks + ak < w; // past the panel's width the clamped column was re-read: its products are masked
            const double m0 = in ? y0[ks] : 0.0, m1 = in ? y1[ks] : 0.0;
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(m0, x0[ks], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(m0, x1[ks], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(m1, x0[ks], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(m1, x1[ks], acc[1][1], 0, 0, 0);
        }
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int row = ia + 16 * mi + ar;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = ja + 16 * nj + ak + 4 * r;
                    if (col < colEnd && row < N && row >= col) F[row + (long long)N * col] = old[nj][mi][r] - acc[nj][mi][r];
                }
            }
        return;
    }
    // ---- role B: bring panel kb1 up to date with panel kb, factor its pivot block, solve this workgroup's rows.
    // The factored pivot block is never written back into the front (other workgroups of this launch still read the raw
    // one); it goes to the dinv slot and is inverted at the end of the factorisation.
    double* Lp = sm; // Lp[k * LDP + q]  = F(kb1 + q, kb + k): panel-kb rows of the pivot block
    double* A11 = sm + NB * LDP; // A11[c * LDP + q] = pivot block, k-major
    {
        // eight unconditional (clamped) loads in flight at once, selected afterwards: see role A
        constexpr int NLB = NB * NB / WGB;
        double vl[NLB], vd[NLB];
#pragma unroll
        for (int it = 0; it < NLB; ++it) {
            const int e = tid + WGB * it;
            const int k = e >> 5, q = e & 31;
            const double* Fq = F + min(kb1 + q, N - 1);
            vl[it] = Fq[(long long)N * (max(kb, 0) + min(k, max(w, 1) - 1))];
            vd[it] = Fq[(long long)N * (kb1 + min(k, w1 - 1))]; // role B only exists for a non-empty panel: w1 >= 1
        }
#pragma unroll
        for (int it = 0; it < NLB; ++it) {
            const int e = tid + WGB * it;
            const int k = e >> 5, q = e & 31;
            Lp[k * LDP + q] = (k < w && q < w1) ? vl[it] : 0.0;
            A11[k * LDP + q] = (k < w1 && q < w1 && q >= k) ? vd[it] : 0.0;
        }
    }
    __syncthreads();
    if (w > 0 && tid < 192) {
        // A11 -= Lp^T Lp (lower triangle) on the matrix cores: waves 0..2 take the 16 x 16 tiles (0,0), (1,0), (1,1).  As scalar
        // FMAs this was 256 LDS reads per thread -- ~4 k cycles of LDS return traffic on the critical chain of every step.
        const int wv = tid >> 6, l = tid & 63;
        const int ti = wv >= 1, tj = wv == 2; // row tile (q), column tile (c)
        const int ar = l & 15, ak = l >> 4;
        f64x4 acc = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
        for (int ks = 0; ks < NB / 4; ++ks)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Lp[(4 * ks + ak) * LDP + 16 * ti + ar], Lp[(4 * ks + ak) * LDP + 16 * tj + ar], acc, 0, 0, 0);
        // D: column = l & 15 -> c, row = (l >> 4) + 4 r -> q
        const int c = 16 * tj + ar;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = 16 * ti + ak + 4 * r;
            if (q >= c) A11[c * LDP + q] -= acc[r];
        }
    }
    __syncthreads();
    double* Xs = sm + 2 * NB * LDP + NB; // Xs[c * LDX + r] = X(r, c), X = L11^-1 (written by the pivot wave)
    // The wave that takes the pivot block rotates with the workgroup index: with the same wave of every workgroup doing it,
    // the pivot chains of all workgroups resident on a CU shared one SIMD while the other three idled behind them.
    const int wv = ((tid >> 6) - wg) & 3; // 0..2: row waves, 3: pivot wave
    const int l = tid & 63, lo = l & 15, hi = l >> 4;
    const int Rw = kb1 + d.z + 16 * MT_B * wv; // first row of this wave
    const bool rowWave = wv < ROW_WAVES_B && Rw < N;
    // Everything the rows need is a product: X_rows = (raw - P_kb Lp^T) L11^-T.  It is formed transposed, tile by tile of 16 rows:
    //   D1^T(k, m) = raw(m, k) - sum_j Lp(k, j) P(m, j)      A = -Lp (LDS), B = rows of panel kb straight from the front
    //   X^T(n, m)  = sum_k Linv(n, k) D1^T(k, m)              A = Linv (LDS), B = D1^T as it sits in the accumulators
    // (accumulator register i holds row (l >> 4) + 4 i, which is the B-operand row of k-step i), and the result lands as 16
    // consecutive rows m per column n: 128-byte stores.  v_mfma_f64_16x16x4_f64: A[l & 15][l >> 4], B[l >> 4][l & 15],
    // D row = (l >> 4) + 4 reg, col = l & 15.
    f64x4 d1t[MT_B][2];
    if (wv == 3) {
        // pivot wave (alone on its SIMD): Cholesky of the 32 x 32 block and its inverse while the row waves fetch and update
        // (the pivot chain is what every other wave of the step ends up waiting for: it gets issue priority on its SIMD)
        __builtin_amdgcn_s_setprio(3);
        if (wave_potrf_inv32_mfma(A11, LDP, w1, l, Xs)) atomicOr(flag, 1);
        __builtin_amdgcn_s_setprio(0);
    }
    else if (rowWave) {
        // every load of the wave (raw rows of the panel and its rows of panel kb, all 16-row tiles) is issued first: tile by
        // tile the compiler waited out two memory round trips per tile, and the row waves ended up as late as the pivot wave
        double pv[MT_B][NB / 4];
#pragma unroll
        for (int mt = 0; mt < MT_B; ++mt) {
            const double* Fr = F + min(Rw + 16 * mt + lo, N - 1);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = 16 * kt + hi + 4 * i;
                    const double v = Fr[(long long)N * (kb1 + min(k, max(w1, 1) - 1))];
                    d1t[mt][kt][i] = (k < w1) ? v : 0.0;
                }
#pragma unroll
            for (int ks = 0; ks < NB / 4; ++ks) pv[mt][ks] = Fr[(long long)N * min(max(kb, 0) + 4 * ks + hi, N - 1)]; // unused when w == 0
        }
        __builtin_amdgcn_sched_barrier(0);
        if (w > 0) { // wave-uniform; a panel that has a successor is always full (w == NB)
#pragma unroll
            for (int mt = 0; mt < MT_B; ++mt)
#pragma unroll
                for (int ks = 0; ks < NB / 4; ++ks) {
                    d1t[mt][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lp[(4 * ks + hi) * LDP + lo], pv[mt][ks], d1t[mt][0], 0, 0, 0);
                    d1t[mt][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lp[(4 * ks + hi) * LDP + 16 + lo], pv[mt][ks], d1t[mt][1], 0, 0, 0);
                }
        }
    }
    __syncthreads();
    if (rowWave) {
#pragma unroll
        for (int mt = 0; mt < MT_B; ++mt) {
            f64x4 x0 = { 0.0, 0.0, 0.0, 0.0 }, x1 = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int k = 4 * ks + hi;
                x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Xs[k * LDX + lo], d1t[mt][0][ks], x0, 0, 0, 0); // Linv(n, k), n < 16: k < 16 only
                x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Xs[k * LDX + 16 + lo], d1t[mt][0][ks], x1, 0, 0, 0);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int k = 16 + 4 * ks + hi;
                x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Xs[k * LDX + 16 + lo], d1t[mt][1][ks], x1, 0, 0, 0);
            }
            const int row = Rw + 16 * mt + lo;
            if (row >= kb1 + w1 && row < N) {
                double* out = F + row + (long long)N * kb1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n0 = hi + 4 * i, n1 = 16 + hi + 4 * i;
                    if (n0 < w1) out[(long long)N * n0] = x0[i];
                    if (n1 < w1) out[(long long)N * n1] = x1[i];
                }
            }
        }
    }
    if (d.z == 0) {
        // the inverse of the pivot block goes straight to its dinv slot (column-major, identity-padded): the solves multiply by it
        double* slot = dinv + ((long long)d.x + kb1 / NB) * (NB * NB);
        for (int e = tid; e < NB * NB; e += WGB) slot[e] = Xs[(e >> 5) * LDX + (e & 31)];
    }
}

template <bool TOP>
__global__ __launch_bounds__(WGB, MF_STEP_OCC) void k_big_step(const int4* __restrict__ desc, TreeView tv, double* __restrict__ fronts,
    double* __restrict__ dinv, int* __restrict__ flag, XinvView xv)
{
    const int wg = blockIdx.x;
    step_work<TOP>(wg, desc[2 * wg], desc[2 * wg + 1], tv, fronts, dinv, flag, xv);
}

// ---- explicit inverses of the factor triangles of the widest fronts ----------------------------------------------------
// The nc x nc triangle of a top-level front is swept by ONE workgroup in the blocked substitution above: 3.2 MB through a
// single CU for the root of a 45 K-node sheet, ~140 us per direction, and the top five levels make up half of the solve.
// For fronts with nc >= XINV_MIN_NC the factorisation therefore also forms X = L11^-1 (recursive doubling on the 32 x 32
// block inverses: X21 = -X22 (L21 X11), two batched GEMM launches per doubling), and the sweeps become two matrix-vector
// products spread over many workgroups.  Extra work: ~nc^3 / 3 flops per front, 4 % of the factorisation.
// X is column-major with leading dimension nc; only its lower triangle is ever read.

// one workgroup per diagonal block: the 32 x 32 inverse goes from its dinv slot into X.  desc = (front, block, 0, 0)
__global__ __launch_bounds__(256) void k_xinv_init(const int4* __restrict__ desc, TreeView tv, XinvView xv, const double* __restrict__ dinv)
{
    const int4 d = desc[blockIdx.x];
    const int s = d.x, b = d.y;
    const int nc = frontNc(tv, s);
    const double* blk = dinv + (tv.dinvOff[s] + b) * (NB * NB); // blk[c * 32 + r] = X(r, c), identity-padded
    double* X = xv.X + xv.xOff[s];
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        const int c = e >> 5, r = e & 31;
        const int R = NB * b + r, C = NB * b + c;
        if (R < nc && C < nc) X[R + (long long)nc * C] = blk[e];
    }
}

// One workgroup per 32 x 32 output tile; its four waves split the k range (batch b of 16 goes to wave b mod 4, partial sums
// combined through LDS in a fixed order) and keep the next batch of operands in flight while the matrix cores work on the
// current one.  Operands come straight from HBM / L2 in the MFMA layout (the matrices are a few MB).  The product is formed
// transposed, D(c, r), so that the 16 lanes of an accumulator row store 16 consecutive rows r of one column.
// d0 = (front, r0, c0, mode), d1 = (rEnd, cEnd, kBeg, kEnd)
//   mode 1: T(r, c)  =   sum_k L(r, k) X(k, c),  k >= c     (X11 lower triangular)
//   mode 2: X(r, c)  = - sum_k X(r, k) T(k, c),  k <= r     (X22 lower triangular)
__global__ __launch_bounds__(256) void k_xinv_gemm(const int4* __restrict__ desc, TreeView tv, XinvView xv, const double* __restrict__ fronts)
{
    __shared__ double red[4][4][256]; // [wave][16 x 16 tile][D layout: 64 lanes x 4]
    const int4 d0 = desc[2 * blockIdx.x], d1 = desc[2 * blockIdx.x + 1];
    const int s = d0.x, r0 = d0.y, c0 = d0.z, mode = d0.w;
    const int rEnd = d1.x, cEnd = d1.y, kEnd = d1.w;
    const int N = frontN(tv, s), nc = frontNc(tv, s);
    const double* F = fronts + tv.frontOff[s];
    double* X = xv.X + xv.xOff[s];
    double* T = xv.T + xv.xOff[s];
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, lo = l & 15, hi = l >> 4;
    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{ 0.0, 0.0, 0.0, 0.0 };
    const int cc[2] = { c0 + lo, c0 + 16 + lo };
    const int rr[2] = { r0 + lo, r0 + 16 + lo };
    const int rrc[2] = { min(rr[0], nc - 1), min(rr[1], nc - 1) };
    const int ccc[2] = { min(cc[0], nc - 1), min(cc[1], nc - 1) };
    // first k that can contribute: the triangular operand is zero before it
    const int kBeg = (mode == 1) ? max(d1.z, c0) : d1.z;
    const int kStop = (mode == 2) ? min(kEnd, r0 + 32) : kEnd;
    // the two operand matrices of this mode: Bop(k, c) -> MFMA A operand, Aop(r, k) -> MFMA B operand
    const double* Bm = (mode == 1) ? X : T;
    const double* Am = (mode == 1) ? F : X;
    const long long ldA = (mode == 1) ? N : nc;
    // Two operand sets, ping-pong: the loads of the next batch are issued (unconditionally, clamped) before the products of the
    // current one, and masked only when they are used -- selecting at load time, copying a "next" set or fetching behind a
    // branch each make the compiler wait for the newest load before the products (see k_big_schur).
    auto fetch = [&](int k0, double (&ra)[2][4], double (&rb)[2][4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kc = min(k0 + 4 * ks + hi, nc - 1);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                ra[q][ks] = Bm[kc + (long long)nc * ccc[q]];
                rb[q][ks] = Am[rrc[q] + ldA * kc];
            }
        }
    };
    auto mult = [&](int k0, const double (&ra)[2][4], const double (&rb)[2][4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = k0 + 4 * ks + hi;
            const bool kin = k < kStop;
            double ma[2], mb[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bool tri = (mode == 1) ? (k >= cc[q]) : (k <= rr[q]);
                ma[q] = (kin && cc[q] < cEnd && (mode == 2 || tri)) ? ra[q][ks] : 0.0;
                mb[q] = (kin && rr[q] < rEnd && (mode == 1 || tri)) ? rb[q][ks] : 0.0;
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ma[a], mb[b], acc[a][b], 0, 0, 0);
        }
    };
    double va[2][4], vb[2][4], na[2][4], nb[2][4];
    int k0 = kBeg + 16 * wv;
    fetch(k0, va, vb);
    for (; k0 < kStop; k0 += 128) {
        fetch(k0 + 64, na, nb);
        __builtin_amdgcn_sched_barrier(0);
        mult(k0, va, vb);
        __builtin_amdgcn_sched_barrier(0);
        fetch(k0 + 128, va, vb);
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + 64 < kStop) mult(k0 + 64, na, nb);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wv][2 * a + b][64 * i + l] = acc[a][b][i];
    __syncthreads();
    // wave q finishes 16 x 16 tile q = 2 a + b
    const int a = wv >> 1, b = wv & 1;
    double* out = (mode == 1) ? T : X;
    const double sgn = (mode == 1) ? 1.0 : -1.0;
    const int r = r0 + 16 * b + lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + 16 * a + hi + 4 * i;
        const double v = ((red[0][wv][64 * i + l] + red[1][wv][64 * i + l]) + red[2][wv][64 * i + l]) + red[3][wv][64 * i + l];
        if (r < rEnd && c < cEnd) out[r + (long long)nc * c] = sgn * v;
    }
}

} // namespace

void MfNumeric::setup(const MfSymbolic& sym, hipStream_t stream, const int* ia_dev, const int* ja_dev, long long nnzPattern)
{
    if (side_) HIP_CHECK(hipStreamSynchronize(side_)); // buffers are about to be replaced
    const bool timeIt = std::getenv("IPCGPU_MF_SETUP_TIMES") != nullptr;
    auto tSetup = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timeIt) return;
        HIP_CHECK(hipStreamSynchronize(stream));
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "mf setup %-28s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tSetup).count());
        tSetup = now;
    };
    sym_ = &sym;
    stream_ = stream;
    ns_ = sym.ns;
    nLevels_ = (int)sym.levelPtr.size() - 1;
    fronts_.ensure((size_t)sym.frontOff[ns_]);
    fronts_.zeroN((size_t)sym.frontOff[ns_], stream); // once per analysis: the numeric phase writes every entry it reads, this only keeps never-read padding finite
    w_.alloc((size_t)sym.wOff[ns_]);
    yperm_.alloc((size_t)sym.n);
    bperm_.alloc((size_t)sym.n);
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
        std::vector<long long> di(ns_ + 1, 0);
        for (int s = 0; s < ns_; ++s) di[s + 1] = di[s] + (sym.nc(s) + NB - 1) / NB;
        dinvOff_.upload(di, stream);
        hDinvOff_ = di;
        nDiagBlocks_ = di[ns_];
        dinv_.alloc((size_t)di[ns_] * NB * NB);
    }
    lap("buffers + tree uploads");
    flag_.alloc(1);
    hflag_.alloc(4);
    if (!side_) {
        HIP_CHECK(hipStreamCreateWithFlags(&side_, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&evSide_, hipEventDisableTiming));
    }
    if (!fwd_) {
        HIP_CHECK(hipStreamCreateWithFlags(&fwd_, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&evRhs_, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&evFwdDone_, hipEventDisableTiming));
    }
    while ((int)evFactLevel_.size() < nLevels_) {
        hipEvent_t e;
        HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        evFactLevel_.push_back(e);
    }
    while ((int)evLevel_.size() < nLevels_) {
        hipEvent_t e;
        HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        evLevel_.push_back(e);
        HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        evInvDone_.push_back(e);
    }
    sidePending_ = false;

    // A front whose nc own columns (plus the index maps of its children) fit into LDS takes the fused single-workgroup path;
    // the others go through the level-batched multi-workgroup kernels.
    size_t fusedLds = 64 * 1024;
    // Fixed since round 6 (each was an environment switch while it was being measured; the sweeps are profiles/r05_knob_sweep*.txt, r05_two_level_blocking_ab.txt,
    // r03r_schur_tile_ab.txt, r04_nd_leaf_size_ab.txt): levels with >= 512 Schur tiles of 32 x 32 take the 64 x 64 kernel (schur64Min_); levels whose step launches
    // move >= 48 MB of own columns factor them in outer blocks of 256 columns (bulkMinMB_, bulkBlock_: members with these defaults -- the only two a caller can set,
    // ipcgpu_linsys_set_tuning, because no mesh of the test suite reaches 48 MB and the path has to be forced to be tested; swept at 375 K nodes: 4, 16, 64 MB the
    // same, factorisation 17.58 -> 16.9 ms; block 128 the same, 512 half the gain); 64 KB of LDS per fused front.
    auto ldsOf = [&](int s) {
        const size_t kids = (size_t)(sym.childPtr[s + 1] - sym.childPtr[s]);
        return ((size_t)sym.nc(s) * sym.N(s) + 64) * sizeof(double) + kids * sym.N(s) * sizeof(int);
    };
    const int ntSmallN = 0, ntBigN = 200;
    // ... unless its level has fronts of the second kind anyway and only a few of the first (round 5): the single-workgroup kernel of such a level is a launch of
    // its own IN FRONT of the level's batched kernels -- 57 us for the 93 widest fused fronts of level 4 of a 45 K-node sheet, one workgroup each at the limit of
    // what LDS holds -- while as members of the batched launches the same fronts cost next to nothing (those launches are latency-bound and far from full).
    // (profiles/r05_mixed_levels_ab_and_p2p_bytes.txt)
    std::vector<char> fusedFront(ns_, 0);
    {
        const bool mixedOk = false;
        std::vector<int> nFit(nLevels_, 0), nBigL(nLevels_, 0);
        for (int s = 0; s < ns_; ++s) {
            fusedFront[s] = sym.childPtr[s + 1] - sym.childPtr[s] <= FUSED_MAX_KIDS && ldsOf(s) <= fusedLds;
            (fusedFront[s] ? nFit : nBigL)[sym.level[s]]++;
        }
        if (!mixedOk)
            for (int s = 0; s < ns_; ++s) {
                const int l = sym.level[s];
                if (fusedFront[s] && nBigL[l] > 0 && nFit[l] <= std::max(64, nBigL[l])) fusedFront[s] = 0;
            }
    }
    auto isFused = [&](int s) { return fusedFront[s] != 0; };
    // explicit triangle inverses (see k_xinv_*): fronts of the multi-workgroup path with nc >= xinvMin
    const int xinvMin = 192;
    xinvBorder_ = true; // the inverse grows by bordering inside the step launches (step_border); false = recursive doubling on the side stream, as before round 4 (profiles/r05_permlane_and_border_ab.txt)
#ifndef MF_BORDER_MAX_NC
#define MF_BORDER_MAX_NC 1536 // (1024 until round 6: the 1 440-column root of the two-sheet contact stack keeps 0.17 ms of doubling rounds behind its factorisation, profiles/r06_border_max_nc_ab.txt)
#endif
    const int borderMaxNc = MF_BORDER_MAX_NC; // wider separators (a root of 2 600 columns at 1.12 M tets) keep the recursive doubling: a bordering workgroup is as long as the
                            // front is wide, and at that width it stretches every step launch (measured at mat433: factorisation 20.0 -> 20.8 ms)
    auto hasXinv = [&](int s) { return !isFused(s) && sym.nc(s) >= xinvMin; };
    auto hasBorder = [&](int s) { return xinvBorder_ && hasXinv(s) && sym.nc(s) <= borderMaxNc; };
    // ---- multi-GPU: cut the assembly tree below its top separators (see mf_numeric.h)
    owner_.assign(ns_, rank_);
    exec_.assign(ns_, rank_);
    sharedFlops_ = 0.0;
    if (world_ > 1) {
        if (world_ > 64) throw StateError("the sharded solver supports at most 64 ranks");
        sharedFlops_ = mf_assign_owners(sym, world_, owner_); // host logic, shared with the CPU tests of the protocol (mf_symbolic.cpp)
        mf_assign_executors(sym, owner_, exec_, group_);
        // exchange lists, by level (mf_exchange_plan: fronts whose parent another rank executes, solution segments of the fronts above the cut).  Every
        // rank computes the same staging layout; it packs what it sends and unpacks what it receives.
        std::vector<MfExchangeLevel> plan;
        mf_exchange_plan(sym, owner_, exec_, group_, rank_, world_, plan);
        xchg_.assign(nLevels_, Xchg());
        std::vector<int4> xd;
        long long maxCount = 1;
        for (const MfExchangeLevel& E : plan) {
            maxCount = std::max(maxCount, E.count + E.countW);
            // the device descriptor of an update vector carries its offset (matrix area + offW) in ONE 32-bit word (k_xchg_w), the matrices' in two
            if ((long long)E.count + (long long)E.countW > (long long)INT_MAX)
                throw StateError("solver exchange: a level's staging area exceeds 2^31 doubles (the update-vector offsets are 32-bit)");
        }
        xchgBuf_.ensure((size_t)maxCount + 1);
        for (int l = 0; l < nLevels_; ++l) {
            Xchg& X = xchg_[l];
            const MfExchangeLevel& E = plan[l];
            X.count = E.count;
            X.countW = E.countW;
            auto emit = [&](const std::vector<MfExchangeItem>& items, Range& R, int sendFlag) {
                R.off = (int)xd.size();
                for (const MfExchangeItem& it : items) {
                    xd.push_back(make_int4(it.front, (int)(unsigned)(it.off & 0xffffffffLL), (int)(it.off >> 32), (int)(E.count + it.offW))); // vectors sit behind the matrices
                    const long long m = sym.N(it.front) - sym.nc(it.front);
                    X.opsM.push_back(P2POp{ xchgBuf_.p + it.off, m * (m + 1) / 2, it.peer, sendFlag });
                    X.opsW.push_back(P2POp{ xchgBuf_.p + E.count + it.offW, m, it.peer, sendFlag });
                }
                R.cnt = (int)xd.size() - R.off;
            };
            emit(E.send, X.pack, 1);
            emit(E.recv, X.unpack, 0);
            for (const MfExchangeItem& it : E.xsSend) X.opsX.push_back(P2POp{ xsol_.p + 3 * (long long)sym.firstNode[it.front], (long long)sym.nc(it.front), it.peer, 1 });
            for (const MfExchangeItem& it : E.xsRecv) X.opsX.push_back(P2POp{ xsol_.p + 3 * (long long)sym.firstNode[it.front], (long long)sym.nc(it.front), it.peer, 0 });
        }
        if (xd.empty()) xd.push_back(make_int4(0, 0, 0, 0));
        xchgDesc_.upload(xd.data(), xd.size(), stream);
        std::vector<int> ne(std::max(sym.nn, 1), 0);
        for (int s = 0; s < ns_; ++s)
            for (int v = sym.firstNode[s]; v < sym.firstNode[s + 1]; ++v) ne[v] = exec_[s];
        nodeExec_.upload(ne, stream);
    }
    auto mine = [&](int s) { return world_ == 1 || exec_[s] == rank_; };
    // The fronts of every level in the order the plans below use them (heaviest first, so that the tail of a level is made of short jobs), the levels whose
    // Schur kernel gathers the update block itself (k_big_schur64_ea: their extend-add only writes own columns), and the numbering of the extend-add tiles
    // (64 x 64, lower triangle, front after front): the entries of A are sorted by the tile they land in, because the extend-add kernel adds them (round 5).
    std::vector<std::vector<int>> smallByLevel(nLevels_), bigByLevel(nLevels_);
    std::vector<char> levelFuseEA(nLevels_, 0);
    std::vector<int> eaTileBase(ns_, -1), eaColTiles(ns_, 0); // first tile of a front; tiles kept per tile row: min(ti + 1, eaColTiles)
    int nEaTiles = 0;
    for (int l = 0; l < nLevels_; ++l) {
        std::vector<int>&small = smallByLevel[l], &big = bigByLevel[l];
        for (int i = sym.levelPtr[l]; i < sym.levelPtr[l + 1]; ++i) {
            const int s = sym.levelFronts[i];
            if (!mine(s)) continue; // factorised and solved by the rank that executes it
            (isFused(s) ? small : big).push_back(s);
        }
        std::sort(small.begin(), small.end(), [&](int a, int b) { return sym.N(a) > sym.N(b) || (sym.N(a) == sym.N(b) && a < b); });
        std::sort(big.begin(), big.end(), [&](int a, int b) { return sym.nc(a) > sym.nc(b) || (sym.nc(a) == sym.nc(b) && a < b); });
        long long tiles32 = 0;
        for (int s : big) {
            const long long nt = (sym.N(s) - sym.nc(s) + TQ - 1) / TQ;
            tiles32 += nt * (nt + 1) / 2;
        }
        levelFuseEA[l] = tiles32 >= schur64Min_;
        for (int s : big) {
            const int nt = (sym.N(s) + TS - 1) / TS;
            eaColTiles[s] = levelFuseEA[l] ? (sym.nc(s) + TS - 1) / TS : nt;
            eaTileBase[s] = nEaTiles;
            for (int ti = 0; ti < nt; ++ti) nEaTiles += std::min(ti + 1, eaColTiles[s]);
        }
    }
    auto eaTileOf = [&](int s, int ti, int tj) { // index of tile (ti, tj) of front s among the extend-add tiles
        const int c = eaColTiles[s];
        return eaTileBase[s] + (ti <= c ? ti * (ti + 1) / 2 : c * (c + 1) / 2 + (ti - c) * c) + tj;
    };
    // entries of A grouped by where they go: (source index, offset inside the LDS panel) for the single-workgroup fronts, per front; (source index, offset in
    // the front buffer) for the others, per extend-add tile -- computed and sorted on the device from the pattern in HBM (k_entry_dst, k_scan_exclusive,
    // k_entry_scatter above).  What comes back to the host: the bucket starts (the packed descriptors of the single-workgroup fronts carry their entry range).
    {
        const size_t nnz = (size_t)nnzPattern;
        const int nBuckets = ns_ + nEaTiles;
        std::vector<int4> info(std::max(ns_, 1));
        for (int s = 0; s < ns_; ++s) info[s] = make_int4(!mine(s) ? -1 : (isFused(s) ? 0 : 1), eaTileBase[s], eaColTiles[s], 0);
        std::vector<int> nodeFront(std::max(sym.nn, 1));
        for (int s = 0; s < ns_; ++s)
            for (int v = sym.firstNode[s]; v < sym.firstNode[s + 1]; ++v) nodeFront[v] = s;
        frontInfo_.upload(info.data(), info.size(), stream);
        nodeFront_.upload(nodeFront, stream);
        entryDst_.ensure(nnz + 1);
        entryBucket_.ensure(nnz + 1);
        bucketHist_.ensure(2 * (size_t)nBuckets + 2); // counts, then the tickets of the scatter
        bucketStart_.ensure((size_t)nBuckets + 2);
        bucketHist_.zeroN(2 * (size_t)nBuckets + 2, stream);
        EntryView ev{ ia_dev, ja_dev, sym.n, ns_, newOf_.p, nodeFront_.p, firstNode_.p, idxPtr_.p, idx_.p, frontOff_.p, frontInfo_.p };
        hipLaunchKernelGGL(k_entry_dst, dim3((sym.n + 3) / 4), dim3(256), 0, stream, ev, entryDst_.p, entryBucket_.p, bucketHist_.p);
        hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, stream, nBuckets, bucketHist_.p, bucketStart_.p);
        std::vector<int> start((size_t)nBuckets + 1);
        bucketStart_.download(start.data(), start.size(), stream); // (synchronises)
        const size_t nFused = (size_t)start[ns_], nBig = (size_t)(start[nBuckets] - start[ns_]);
        aPtrHost_.assign(start.begin(), start.begin() + ns_ + 1);
        nFusedA_ = (int)nFused;
        aPerm_.ensure(std::max<size_t>(nFused, 1));
        aSrc_.ensure(nFused + 1);
        aLoc_.ensure(nFused + 1);
        bigASrc_.ensure(nBig + 1);
        bigADst_.ensure(nBig + 1);
        if (nnz)
            hipLaunchKernelGGL(k_entry_scatter, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, stream, (int)nnz, ns_, entryDst_.p, entryBucket_.p, bucketStart_.p,
                bucketHist_.p + nBuckets + 1, frontOff_.p, aSrc_.p, aLoc_.p, bigASrc_.p, bigADst_.p);
        std::vector<int> eaPtr((size_t)nEaTiles + 1);
        for (int t = 0; t <= nEaTiles; ++t) eaPtr[t] = start[ns_ + t] - start[ns_];
        eaAPtr_.upload(eaPtr, stream); // per extend-add tile: its range in bigASrc_ / bigADst_
    }
    lap("A-entry lists");
    plan_.assign(nLevels_, LevelPlan());
    std::vector<int> smallList, bigList;
    std::vector<int4> ea;
    std::vector<int> bigFd; // packed records of the fronts of the multi-workgroup path (k_extend_add)
    std::vector<int> eaRecOf(ns_, -1); // front -> its first record
    std::vector<int4> desc;
    size_t maxSmallLds = 0, maxSolveLds = 0, maxTriLds = 0, maxBwdLds = 0;
    for (int l = 0; l < nLevels_; ++l) {
        LevelPlan& P = plan_[l];
        const std::vector<int>&small = smallByLevel[l], &big = bigByLevel[l];
        P.small.off = (int)smallList.size();
        P.small.cnt = (int)small.size();
        int maxN = 0, maxNc = 0;
        for (int s : small) maxN = std::max(maxN, sym.N(s));
        for (int s : big) maxNc = std::max(maxNc, sym.nc(s));
        smallList.insert(smallList.end(), small.begin(), small.end());
        P.bigFronts.off = (int)bigList.size();
        P.bigFronts.cnt = (int)big.size();
        bigList.insert(bigList.end(), big.begin(), big.end());
        P.smallLds = 0;
        for (int s : small) P.smallLds = std::max(P.smallLds, ldsOf(s));
        // narrow fronts: two waves per workgroup, twice as many workgroups per CU (every phase of the kernel is latency-bound)
        P.smallThreads = (maxN <= ntSmallN) ? 128 : (maxN >= ntBigN ? 512 : 256); // wide fronts: one workgroup per CU anyway (LDS)
        P.solveLds = (size_t)std::max(maxN, 1) * sizeof(double);
        P.triLds = (size_t)std::max(maxNc, 1) * sizeof(double);
        {
            int maxBelow = 1;
            for (int s : big) maxBelow = std::max(maxBelow, sym.N(s) - sym.nc(s));
            P.bwdLds = (size_t)maxBelow * sizeof(double);
            maxBwdLds = std::max(maxBwdLds, P.bwdLds);
        }
        maxSmallLds = std::max(maxSmallLds, P.smallLds);
        maxSolveLds = std::max(maxSolveLds, P.solveLds);
        maxTriLds = std::max(maxTriLds, P.triLds);
        // extend-add descriptors: lower-triangular 64 x 64 tiles of the parent, each pointing at the parent's packed record
        P.ea.off = (int)ea.size();
        P.schur64 = P.fuseEA = levelFuseEA[l] != 0; // (decided above: it thins out the extend-add's tiles and numbers them)
        for (int s : big) { // every lower-triangle tile is written (children sums or zeros): the fronts are never zero-filled
            const int first = (int)(bigFd.size() / FD_STRIDE);
            eaRecOf[s] = first;
            const int nkAll = sym.childPtr[s + 1] - sym.childPtr[s];
            for (int k0 = 0; k0 == 0 || k0 < nkAll; k0 += FUSED_MAX_KIDS) {
                const size_t base = bigFd.size();
                bigFd.resize(base + FD_STRIDE, 0);
                int* d = bigFd.data() + base;
                const long long off = sym.frontOff[s];
                std::memcpy(d, &off, 8);
                d[2] = sym.N(s);
                d[3] = sym.nc(s);
                const int nk = std::min(FUSED_MAX_KIDS, nkAll - k0);
                d[8] = std::max(nk, 0);
                d[9] = (k0 + FUSED_MAX_KIDS < nkAll) ? (int)(base / FD_STRIDE) + 1 : -1;
                for (int q = 0; q < nk; ++q) {
                    const int c = sym.child[sym.childPtr[s] + k0 + q];
                    int* k = d + 16 + 6 * q;
                    const long long coff = sym.frontOff[c];
                    std::memcpy(k, &coff, 8);
                    k[2] = sym.N(c);
                    k[3] = sym.nc(c);
                    k[4] = sym.invPtr[c];
                }
            }
            const int nt = (sym.N(s) + TS - 1) / TS;
            for (int ti = 0; ti < nt; ++ti)
                for (int tj = 0; tj <= ti; ++tj) {
                    if (P.fuseEA && TS * tj >= sym.nc(s)) continue; // a tile of the update block alone: the Schur kernel's
                    if ((int)ea.size() != eaTileOf(s, ti, tj)) throw StateError("internal: extend-add tiles are not numbered in emission order");
                    ea.push_back(make_int4(first, ti, tj, eaTileOf(s, ti, tj)));
                }
        }
        P.ea.cnt = (int)ea.size() - P.ea.off;
        // big-front step descriptors: launch 0 factors panel 0, launch j + 1 applies panel j and factors panel j + 1
        int steps = 0;
        for (int s : big) steps = std::max(steps, (sym.nc(s) + NB - 1) / NB);
        P.step.assign(big.empty() ? 0 : steps + 1, Range());
        P.bulk.assign(P.step.size(), Range());
        // two-level blocking (k_big_bulk) for the fronts of this level?  What a step launch's rank-32 update reads and writes: the own columns of every front, all rows
        const int OBW = bulkBlock_;
        double stepMB = 0.0;
        for (int s : big) stepMB += 8.0e-6 * (double)sym.N(s) * sym.nc(s);
        const bool levelWide = stepMB >= bulkMinMB_;
        for (int j = -1; j < steps && !big.empty(); ++j) {
            Range& R = P.step[j + 1];
            R.off = (int)desc.size();
            for (int s : big) {
                const int N = sym.N(s), nc = sym.nc(s);
                const int kb = j * NB;
                if (j >= 0 && kb >= nc) continue;
                const int w = (j >= 0) ? std::min(NB, nc - kb) : 0;
                const int kb1 = (j >= 0) ? kb + w : 0;
                const int w1 = (kb1 < nc) ? std::min(NB, nc - kb1) : 0;
                const long long foff = sym.frontOff[s];
                const int4 rec2 = make_int4(N, nc, (int)(unsigned)(foff & 0xffffffffll), (int)(unsigned)(foff >> 32));
                // wide fronts (two-level blocking, k_big_bulk): E = the end of the outer block panel kb belongs to; when the next panel opens a new block, the
                // bulk update launched in front of this step has applied panel kb already
                const bool wide = levelWide && nc >= 2 * OBW;
                const int E = (wide && j >= 0) ? std::min(nc, (kb / OBW + 1) * OBW) : nc;
                const bool applied = wide && j >= 0 && kb1 >= E && kb1 < nc;
                if (w1 > 0)
                    for (int r0 = 0; r0 < N - kb1; r0 += ROWS_B) {
                        desc.push_back(make_int4((int)hDinvOff_[s], j < 0 ? -1 : (applied ? -2 - kb1 : kb), r0, -2));
                        desc.push_back(rec2);
                    }
                if (j >= 0 && hasBorder(s)) { // role C: the rows of panel j of X = L11^-1, one workgroup per column tile up to the diagonal block
                    P.stepTop = true;
                    // 16-column tiles (32 wide ones made the late steps of the root 20 us long: one CU per tile, k up to nc); c0 == kb: the diagonal block, one copy
                    for (int c0 = 0; c0 <= kb; c0 += 16) {
                        desc.push_back(make_int4(s, kb, c0, -6));
                        desc.push_back(rec2);
                    }
                }
                if (j >= 0 && !applied) {
                    // trailing tiles inside the front's own columns (of a wide front: inside the panel's outer block -- the records carry E in place of nc, which is
                    // all role A reads nc for); the Schur complement (columns >= nc) waits for k_big_schur
                    const int M0 = kb1 + w1;
                    const int ntr = (N - M0 + TS - 1) / TS, ntc = (E - M0 + TS - 1) / TS;
                    const int4 recA = make_int4(N, E, rec2.z, rec2.w);
                    for (int ti = 0; ti < ntr; ++ti)
                        for (int tj = 0; tj <= ti && tj < ntc; ++tj) {
                            desc.push_back(make_int4((int)hDinvOff_[s], kb, ti, tj));
                            desc.push_back(recA);
                        }
                }
            }
            R.cnt = ((int)desc.size() - R.off) / 2; // workgroups: two records each
            // the bulk updates that have to run BEHIND this launch (it factored panel j + 1): of every wide front whose outer block ends with that panel
            Range& U = P.bulk[j + 1];
            U.off = (int)desc.size();
            for (int s : big) {
                const int N = sym.N(s), nc = sym.nc(s);
                const int p0 = (j + 1) * NB, Eb = p0 + NB; // the panel just factored and its end
                if (!levelWide || nc < 2 * OBW || Eb % OBW != 0 || Eb >= nc) continue;
                const long long foff = sym.frontOff[s];
                const int4 rec2 = make_int4(N, nc, (int)(unsigned)(foff & 0xffffffffll), (int)(unsigned)(foff >> 32));
                const int ntr = (N - Eb + TQ64 - 1) / TQ64, ntc = (nc - Eb + TQ64 - 1) / TQ64;
                for (int ti = 0; ti < ntr; ++ti)
                    for (int tj = 0; tj <= ti && tj < ntc; ++tj) {
                        desc.push_back(make_int4(OBW, Eb - OBW, ti, tj));
                        desc.push_back(rec2);
                    }
            }
            U.cnt = ((int)desc.size() - U.off) / 2;
        }
        // Schur complement: one pass behind the chain (k_big_schur / k_big_schur64 / k_big_schur64_ea).
        // XCD-aware order (round 5): workgroup b of a launch runs on XCD b % 8 (observed, MI355X_MICROARCH.md; a speed assumption only -- any placement is
        // correct) and every XCD has its own 4 MB L2.  In front-after-front order the tiles of one front land on all eight XCDs, so each L2 sees the factor
        // panels L21 of ALL fronts of the level (25 MB on the 64-front level of a 45 K-node sheet) and every operand load is an L2 miss.  Here the tile rows of
        // a front form groups of about total / 8 tiles, the groups are dealt to eight bins (largest first onto the least loaded bin) and slot b of the
        // launch takes the next tile of bin b % 8: an XCD works through whole fronts and reads their panels from memory once.
        P.schur.off = (int)desc.size();
        {
            const int TQl = P.schur64 ? TQ64 : TQ;
            struct Tile {
                int4 a, b;
            };
            std::vector<std::vector<Tile>> groups;
            long long total = 0;
            for (int s : big) {
                const long long nt = (sym.N(s) - sym.nc(s) + TQl - 1) / TQl;
                total += nt * (nt + 1) / 2;
            }
            const long long target = std::max<long long>(1, (total + XCDS - 1) / XCDS);
            for (int s : big) {
                const int nt = (sym.N(s) - sym.nc(s) + TQl - 1) / TQl;
                const long long foff = sym.frontOff[s];
                const int4 rec2 = make_int4(sym.N(s), sym.nc(s), (int)(unsigned)(foff & 0xffffffffll), (int)(unsigned)(foff >> 32));
                groups.emplace_back();
                for (int ti = 0; ti < nt; ++ti) {
                    if (xcdOrder_ && (long long)groups.back().size() + ti + 1 > target && !groups.back().empty()) groups.emplace_back(); // next row range of a front too large for one bin
                    for (int tj = 0; tj <= ti; ++tj) groups.back().push_back(Tile{ make_int4(s, ti, tj, P.fuseEA ? eaRecOf[s] : 0), rec2 });
                }
            }
            if (!xcdOrder_) {
                for (const auto& g : groups)
                    for (const Tile& t : g) {
                        desc.push_back(t.a);
                        desc.push_back(t.b);
                    }
            }
            else {
                std::vector<int> order(groups.size());
                for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
                std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return groups[a].size() > groups[b].size(); });
                std::vector<std::vector<Tile>> bins(XCDS);
                for (int g : order) {
                    int best = 0;
                    for (int x = 1; x < XCDS; ++x)
                        if (bins[x].size() < bins[best].size()) best = x;
                    bins[best].insert(bins[best].end(), groups[g].begin(), groups[g].end());
                }
                std::vector<size_t> head(XCDS, 0), tail(XCDS);
                for (int x = 0; x < XCDS; ++x) tail[x] = bins[x].size();
                for (long long left = total; left > 0;)
                    for (int x = 0; x < XCDS && left > 0; ++x, --left) {
                        int from = x;
                        if (head[x] >= tail[x]) { // this bin has run dry: take from the END of the fullest one (its head keeps its order)
                            for (int y = 0; y < XCDS; ++y)
                                if (tail[y] - head[y] > tail[from] - head[from]) from = y;
                        }
                        const Tile& t = (from == x) ? bins[x][head[x]++] : bins[from][--tail[from]];
                        desc.push_back(t.a);
                        desc.push_back(t.b);
                    }
            }
        }
        P.schur.cnt = ((int)desc.size() - P.schur.off) / 2; // workgroups: two records each
        P.fwdRect.off = (int)desc.size();
        for (int s : big)
            for (int r0 = 0; r0 < sym.N(s) - sym.nc(s); r0 += MV_ROWS) desc.push_back(make_int4(s, r0, 0, 0));
        P.fwdRect.cnt = (int)desc.size() - P.fwdRect.off;
        P.bwdInit.off = (int)desc.size();
        for (int s : big)
            if (sym.N(s) > sym.nc(s))
                for (int c0 = 0; c0 < sym.nc(s); c0 += 16) desc.push_back(make_int4(s, c0, 0, 0));
        P.bwdInit.cnt = (int)desc.size() - P.bwdInit.off;
    }
    lap("level plans");
    {
        // packed descriptors of the fused fronts, in launch order
        std::vector<int> fd((std::max<size_t>(smallList.size(), 1)) * FD_STRIDE, 0);
        std::vector<long long> di(ns_ + 1, 0);
        for (int s = 0; s < ns_; ++s) di[s + 1] = di[s] + (sym.nc(s) + NB - 1) / NB;
        for (size_t i = 0; i < smallList.size(); ++i) {
            const int s = smallList[i];
            int* d = fd.data() + i * FD_STRIDE;
            const long long off = sym.frontOff[s];
            std::memcpy(d, &off, 8);
            d[2] = sym.N(s);
            d[3] = sym.nc(s);
            std::memcpy(d + 4, &di[s], 8);
            d[6] = aPtrHost_[s];
            d[7] = aPtrHost_[s + 1];
            const int nk = sym.childPtr[s + 1] - sym.childPtr[s];
            d[8] = nk;
            for (int q = 0; q < nk; ++q) {
                const int c = sym.child[sym.childPtr[s] + q];
                int* k = d + 16 + 6 * q;
                const long long coff = sym.frontOff[c];
                std::memcpy(k, &coff, 8);
                k[2] = sym.N(c);
                k[3] = sym.nc(c);
                k[4] = sym.invPtr[c];
            }
        }
        fdesc_.uploadGrow(fd, stream);
    }
    lap("fused descriptors");
    {
        std::vector<long long> xOff(ns_, -1);
        long long xTot = 0;
        std::vector<int4> xd; // all descriptors of the inverse machinery
        std::vector<int> invFronts;
        size_t maxInvNc = 0;
        for (int l = 0; l < nLevels_; ++l)
            for (int i = plan_[l].bigFronts.off; i < plan_[l].bigFronts.off + plan_[l].bigFronts.cnt; ++i) {
                const int s = bigList[i];
                if (!hasXinv(s)) continue;
                xOff[s] = xTot;
                xTot += (long long)sym.nc(s) * sym.nc(s);
                invFronts.push_back(s);
                maxInvNc = std::max<size_t>(maxInvNc, sym.nc(s));
            }
        // per level (the inverses of a level are formed on a side stream while the levels above factorise): the diagonal
        // blocks to invert, the copies into X and the doubling rounds
        std::vector<int> blockList;
        std::vector<long long> di(ns_ + 1, 0);
        for (int s = 0; s < ns_; ++s) di[s + 1] = di[s] + (sym.nc(s) + NB - 1) / NB;
        xinvLevel_.assign(nLevels_, XinvLevel());
        for (int l = 0; l < nLevels_; ++l) {
            XinvLevel& XL = xinvLevel_[l];
            std::vector<int> lf;
            for (int s : invFronts)
                if (sym.level[s] == l) lf.push_back(s);
            XL.blocks.off = (int)blockList.size();
            XL.init.off = (int)xd.size();
            lf.erase(std::remove_if(lf.begin(), lf.end(), [&](int s) { return hasBorder(s); }), lf.end()); // the step launches build those inverses (step_border)
            for (int s : lf)
                for (int b = 0; b < (sym.nc(s) + NB - 1) / NB; ++b) {
                    blockList.push_back((int)(di[s] + b));
                    xd.push_back(make_int4(s, b, 0, 0));
                }
            XL.blocks.cnt = (int)blockList.size() - XL.blocks.off;
            XL.init.cnt = (int)xd.size() - XL.init.off;
            if ((xd.size() & 1) != 0) xd.push_back(make_int4(0, 0, 0, 0)); // GEMM descriptors are pairs: keep them pair-aligned
            int lvlMax = 0;
            for (int s : lf) lvlMax = std::max(lvlMax, sym.nc(s));
            for (int sz = NB; sz < lvlMax; sz *= 2) {
                // pairs (A, C) of this doubling: A = [2 p sz, 2 p sz + sz), C = [2 p sz + sz, min(2 p sz + 2 sz, nc))
                Range g1, g2;
                for (int mode = 1; mode <= 2; ++mode) {
                    Range& g = (mode == 1) ? g1 : g2;
                    g.off = (int)xd.size() / 2;
                    for (int s : lf) {
                        const int nc = sym.nc(s);
                        for (int a0 = 0; a0 + sz < nc; a0 += 2 * sz) {
                            const int c0 = a0 + sz, cEnd = std::min(a0 + 2 * sz, nc);
                            for (int r = c0; r < cEnd; r += 32)
                                for (int c = a0; c < a0 + sz; c += 32) {
                                    xd.push_back(make_int4(s, r, c, mode));
                                    // mode 1 sums over the columns of A, mode 2 over the rows of C
                                    xd.push_back(mode == 1 ? make_int4(cEnd, a0 + sz, a0, a0 + sz) : make_int4(cEnd, a0 + sz, c0, cEnd));
                                }
                        }
                    }
                    g.cnt = (int)xd.size() / 2 - g.off;
                }
                XL.rounds.push_back({ g1, g2 });
            }
        }
        // solve: per level the fronts swept by one workgroup (no inverse) and the row / column blocks of the others
        std::vector<int> triList;
        for (int l = 0; l < nLevels_; ++l) {
            LevelPlan& P = plan_[l];
            P.bigTri.off = (int)triList.size();
            size_t triMax = 1;
            std::vector<int4> fw, bw;
            for (int i = P.bigFronts.off; i < P.bigFronts.off + P.bigFronts.cnt; ++i) {
                const int s = bigList[i];
                if (xOff[s] < 0) {
                    triList.push_back(s);
                    triMax = std::max<size_t>(triMax, sym.nc(s));
                    continue;
                }
                for (int r0 = 0; r0 < sym.nc(s); r0 += MV_ROWS) fw.push_back(make_int4(s, r0, 0, 0));
                for (int c0 = 0; c0 < sym.nc(s); c0 += 16) bw.push_back(make_int4(s, c0, 0, 0));
            }
            P.bigTri.cnt = (int)triList.size() - P.bigTri.off;
            P.triLds = triMax * sizeof(double);
            if ((xd.size() & 1) != 0) xd.push_back(make_int4(0, 0, 0, 0));
            P.xinvFwd.off = (int)xd.size();
            xd.insert(xd.end(), fw.begin(), fw.end());
            P.xinvFwd.cnt = (int)fw.size();
            P.xinvBwd.off = (int)xd.size();
            xd.insert(xd.end(), bw.begin(), bw.end());
            P.xinvBwd.cnt = (int)bw.size();
        }
        maxTriLds = 0;
        for (int l = 0; l < nLevels_; ++l) maxTriLds = std::max(maxTriLds, plan_[l].triLds);
        xinvLds_ = std::max<size_t>(maxInvNc, 1) * sizeof(double);
        if (xinvLds_ > 150 * 1024) throw StateError("a separator front is too wide for the inverse-based triangular solve");
        if (triList.empty()) triList.push_back(0);
        triList_.upload(triList, stream);
        if (xd.empty()) xd.push_back(make_int4(0, 0, 0, 0));
        xinvDesc_.upload(xd.data(), xd.size(), stream);
        xinvOff_.upload(xOff.empty() ? std::vector<long long>{ -1 } : xOff, stream);
        xinvX_.alloc((size_t)std::max<long long>(xTot, 1));
        xinvT_.alloc((size_t)std::max<long long>(xTot, 1));
        xinvX_.zero(stream);
        xinvT_.zero(stream);
    }
    lap("inverse plan + buffers");
    if (smallList.empty()) smallList.push_back(0);
    smallList_.upload(smallList, stream);
    if (bigList.empty()) bigList.push_back(0);
    bigList_.upload(bigList, stream);
    if (ea.empty()) ea.push_back(make_int4(0, 0, 0, 0));
    eaDesc_.upload(ea.data(), ea.size(), stream);
    if (bigFd.empty()) bigFd.resize(FD_STRIDE, 0);
    bigFd_.uploadGrow(bigFd, stream);
    if (desc.empty()) desc.push_back(make_int4(0, 0, 0, 0));
    desc_.upload(desc.data(), desc.size(), stream);
    if (maxSmallLds > 48 * 1024)
    {
        HIP_CHECK(hipFuncSetAttribute((const void*)k_front_fused<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxSmallLds));
        HIP_CHECK(hipFuncSetAttribute((const void*)k_front_fused<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxSmallLds));
        HIP_CHECK(hipFuncSetAttribute((const void*)k_front_fused<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxSmallLds));
    }
    configureSweepKernels(maxSolveLds, maxBwdLds, maxTriLds); // the dynamic-LDS limits of the sweep kernels (mf_sweeps.hip)
    HIP_CHECK(hipStreamSynchronize(stream));
    lap("uploads + attributes");
}

void MfNumeric::nodeOwners(std::vector<int>& ownerOfNode) const
{
    if (!sym_) throw StateError("nodeOwners before analyze_pattern");
    const MfSymbolic& sym = *sym_;
    ownerOfNode.assign(sym.nn, -1);
    if (world_ <= 1) return;
    std::vector<int> perm(sym.nn, -1); // owner per permuted node
    for (int s = 0; s < ns_; ++s)
        for (int v = sym.firstNode[s]; v < sym.firstNode[s + 1]; ++v) perm[v] = owner_[s];
    for (int v = 0; v < sym.nn; ++v) ownerOfNode[v] = perm[sym.newOf[v]];
}

MfNumeric::~MfNumeric()
{
    if (side_) (void)hipStreamSynchronize(side_);
    if (fwd_) (void)hipStreamSynchronize(fwd_);
    for (hipEvent_t e : evFactLevel_) (void)hipEventDestroy(e);
    if (evRhs_) (void)hipEventDestroy(evRhs_);
    if (evFwdDone_) (void)hipEventDestroy(evFwdDone_);
    if (fwd_) (void)hipStreamDestroy(fwd_);
    for (hipEvent_t e : evLevel_) (void)hipEventDestroy(e);
    for (hipEvent_t e : evInvDone_) (void)hipEventDestroy(e);
    if (evSide_) (void)hipEventDestroy(evSide_);
    if (side_) (void)hipStreamDestroy(side_);
}

bool MfNumeric::factorize(const double* a_dev)
{
    if (!sym_) throw StateError("factorize before analyze_pattern");
    enqueueFactor(a_dev);
    if (world_ > 1) { // ALWAYS once more at the end: the exchanges only carry the flag of pivots met before them, and a tree of the
        // forest that was not cut (several bodies, world > number of roots) lives on one rank alone -- one double
        allreduceFlag();
    }
    hipLaunchKernelGGL(k_publish_flag, dim3(1), dim3(1), 0, stream_, flag_.p, hflag_.dev); // mapped pinned memory: no blit
    HIP_CHECK(hipStreamSynchronize(stream_));
#ifdef MF_FUSED_PROBE
    {
        static int calls = 0;
        if (++calls == 3) { // a warm one
            std::vector<unsigned long long> h(16 << 16);
            HIP_CHECK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_probe), h.size() * sizeof(unsigned long long)));
            const char* names[7] = { "maps", "children", "A", "panels", "store", "schur", "total" };
            for (int l = 0; l < nLevels_; ++l) {
                const LevelPlan& P = plan_[l];
                if (!P.small.cnt) continue;
                double sum[16] = { 0 }, mx = 0;
                for (int b = 0; b < P.small.cnt; ++b) {
                    const unsigned long long* r = h.data() + 16 * (size_t)((P.small.off + b) & 0xffff);
                    for (int k = 0; k < 6; ++k) sum[k] += (double)(r[k + 1] - r[k]);
                    sum[6] += (double)(r[6] - r[0]);
                    mx = std::max(mx, (double)(r[6] - r[0]));
                    for (int k = 8; k < 16; ++k) sum[k] += (double)r[k];
                }
                const double c = 0.01 / P.small.cnt; // 100 MHz ticks -> us, averaged
                std::fprintf(stderr, "[fused probe] level %d: %d fronts, %d threads, N %.0f nc %.0f kids %.1f |", l, P.small.cnt, P.smallThreads, sum[12] / P.small.cnt, sum[13] / P.small.cnt,
                    sum[14] / P.small.cnt);
                for (int k = 0; k < 7; ++k) std::fprintf(stderr, " %s %.2f", names[k], sum[k] * c);
                std::fprintf(stderr, " us (max %.2f) | inside panels: pivot %.2f rows %.2f trailing %.2f | inside schur (wave 0): gathers %.2f products + stores %.2f\n", mx * 0.01, sum[8] * c, sum[9] * c,
                    sum[10] * c, sum[11] * c, sum[15] * c);
            }
        }
    }
#endif
    return pivotsOk();
}

bool MfNumeric::pivotsOk() const
{
    return hflag_.p[0] == 0;
}

bool MfNumeric::factorizeSolve(const double* a_dev, const double* rhs_dev, double* x_dev, bool wait)
{
    if (!sym_) throw StateError("factorize before analyze_pattern");
    if (world_ > 1 || !fwd_) { // sharded runs keep the two-call sequence
        const bool ok = factorize(a_dev);
        if (ok) solve(rhs_dev, x_dev);
        return ok;
    }
    // the right-hand side is ready on the main stream now: permute it on the forward stream, then level by level behind the factorisation
    HIP_CHECK(hipEventRecord(evRhs_, stream_));
    HIP_CHECK(hipStreamWaitEvent(fwd_, evRhs_, 0));
    enqueuePermuteRhs(rhs_dev, fwd_);
    fwdJoined_ = false;
    enqueueFactor(a_dev, true);
    if (!fwdJoined_) HIP_CHECK(hipEventRecord(evFwdDone_, fwd_));
    hipLaunchKernelGGL(k_publish_flag, dim3(1), dim3(1), 0, stream_, flag_.p, hflag_.dev);
    // the backward sweep follows the root's forward result; enqueued before the flag is looked at (a failed pivot makes x meaningless,
    // the caller falls back to the diagonal preconditioner as after factorize() == false)
    if (!fwdJoined_) HIP_CHECK(hipStreamWaitEvent(stream_, evFwdDone_, 0));
    enqueueBackward(x_dev);
    if (!wait) return true;
    HIP_CHECK(hipStreamSynchronize(stream_));
    return pivotsOk();
}

void MfNumeric::enqueueFactor(const double* a_dev, bool overlapForward)
{
    TreeView tv{ frontOff_.p, idxPtr_.p, firstNode_.p, childPtr_.p, child_.p, invPtr_.p, inv_.p, idx_.p, dinvOff_.p };
    XinvView xvF{ xinvOff_.p, xinvX_.p, xinvT_.p };
    if (sidePending_) HIP_CHECK(hipStreamWaitEvent(stream_, evSide_, 0)); // the side stream still reads the previous factor
    bool sideUsed = false;
    int fwdNext = 0; // first level not yet handed to the forward stream
#ifdef MF_FUSED_PROBE
    {
        const int* base = fdesc_.p;
        HIP_CHECK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_probeBase), &base, sizeof(base), 0, hipMemcpyHostToDevice, stream_));
    }
#endif
    if (nFusedA_) hipLaunchKernelGGL(k_gather_a, dim3((nFusedA_ + 255) / 256), dim3(256), 0, stream_, nFusedA_, aSrc_.p, a_dev, aPerm_.p, flag_.p);
    else flag_.zero(stream_);
    for (int l = 0; l < nLevels_; ++l) {
        const LevelPlan& P = plan_[l];
        if (P.small.cnt)
        {
            const int* fd = fdesc_.p + (size_t)P.small.off * FD_STRIDE;
            if (P.smallThreads == 128)
                hipLaunchKernelGGL(k_front_fused<128>, dim3(P.small.cnt), dim3(128), P.smallLds, stream_, fd, inv_.p, aLoc_.p, aPerm_.p, fronts_.p,
                    dinv_.p, flag_.p);
            else if (P.smallThreads == 512)
                hipLaunchKernelGGL(k_front_fused<512>, dim3(P.small.cnt), dim3(512), P.smallLds, stream_, fd, inv_.p, aLoc_.p, aPerm_.p, fronts_.p,
                    dinv_.p, flag_.p);
            else
                hipLaunchKernelGGL(k_front_fused<256>, dim3(P.small.cnt), dim3(256), P.smallLds, stream_, fd, inv_.p, aLoc_.p, aPerm_.p, fronts_.p,
                    dinv_.p, flag_.p);
        }
        if (P.ea.cnt) {
            hipLaunchKernelGGL(k_extend_add, dim3(P.ea.cnt), dim3(WG), 0, stream_, eaDesc_.p + P.ea.off, bigFd_.p, inv_.p, fronts_.p, P.fuseEA ? 1 : 0, eaAPtr_.p,
                bigASrc_.p, bigADst_.p, a_dev);
        }
        for (const Range& R : P.step) { // one launch per 32-column step, each with look-ahead (k_big_step)
            if (!R.cnt) continue;
            if (P.stepTop) hipLaunchKernelGGL(k_big_step<true>, dim3(R.cnt), dim3(WGB), 0, stream_, desc_.p + R.off, tv, fronts_.p, dinv_.p, flag_.p, xvF);
            else hipLaunchKernelGGL(k_big_step<false>, dim3(R.cnt), dim3(WGB), 0, stream_, desc_.p + R.off, tv, fronts_.p, dinv_.p, flag_.p, xvF);
            const Range& U = P.bulk[&R - P.step.data()]; // wide fronts: the outer block that ended with this launch's panel goes to the own columns behind it
            if (U.cnt) hipLaunchKernelGGL(k_big_bulk, dim3(U.cnt), dim3(WG), 0, stream_, desc_.p + U.off, fronts_.p);
        }
        if (P.schur.cnt) {
            if (P.fuseEA) hipLaunchKernelGGL(k_big_schur64_ea, dim3(P.schur.cnt), dim3(WG), 0, stream_, desc_.p + P.schur.off, bigFd_.p, inv_.p, fronts_.p);
            else if (P.schur64) hipLaunchKernelGGL(k_big_schur64, dim3(P.schur.cnt), dim3(WG), 0, stream_, desc_.p + P.schur.off, fronts_.p);
            else hipLaunchKernelGGL(k_big_schur, dim3(P.schur.cnt), dim3(WG), 0, stream_, desc_.p + P.schur.off, fronts_.p);
        }
        if (world_ > 1) exchangeUpdateMatrices(l); // update matrices whose parent front another rank executes (mf_exchange.hip)
        if (xinvLevel_[l].blocks.cnt) {
            // the factor panels and pivot blocks of this level are final: form the triangle inverses of its fronts beside the
            // latency-bound chain of the levels above
            if (side_) {
                HIP_CHECK(hipEventRecord(evLevel_[l], stream_));
                HIP_CHECK(hipStreamWaitEvent(side_, evLevel_[l], 0));
                enqueueInverses(l, side_);
                HIP_CHECK(hipEventRecord(evInvDone_[l], side_));
                sideUsed = true;
            }
            else enqueueInverses(l, stream_);
        }
        if (overlapForward) {
            // everything the forward sweep of this level reads is final (factor panels, pivot-block inverses; the triangle inverses of the
            // widest fronts follow on the side stream): hand the level to the forward stream, which runs beside the levels above
            if (l == nLevels_ - 1 && !sideUsed) {
                // the last level (the root separator) has nothing to run beside: its forward sweep follows its factorisation on THIS stream, and the
                // hand-over back from the forward stream happens here, where that stream has long been idle, instead of behind the whole factorisation
                // (a cross-stream event wait costs the waiting stream ~20 us on this runtime: it used to sit between the forward and the backward sweep)
                HIP_CHECK(hipEventRecord(evFwdDone_, fwd_));
                HIP_CHECK(hipStreamWaitEvent(stream_, evFwdDone_, 0));
                enqueueForwardLevel(l, stream_);
                fwdJoined_ = true;
            }
            else if (l >= nLevels_ - 2 || (l + 1) % fwdStride_ == 0) {
                // An event recorded on this stream costs the chain a ~7 us bubble (profiles/r04_schur_passes_on_a_side_stream_ab.txt), so the levels are handed
                // over in groups of fwdStride_: the forward stream has the whole rest of the factorisation to catch up, only the level below the root
                // must not wait (the root's own sweep follows on this stream)
                HIP_CHECK(hipEventRecord(evFactLevel_[l], stream_));
                HIP_CHECK(hipStreamWaitEvent(fwd_, evFactLevel_[l], 0));
                for (; fwdNext <= l; ++fwdNext) {
                    if (plan_[fwdNext].xinvFwd.cnt && sideUsed) HIP_CHECK(hipStreamWaitEvent(fwd_, evInvDone_[fwdNext], 0));
                    enqueueForwardLevel(fwdNext, fwd_);
                }
            }
        }
    }
    // The inverses are first needed when the forward solve reaches their level (the root's: at its very end), so the main
    // stream does not wait here: enqueueSolve waits per level, the next factorisation waits before it touches the fronts.
    if (sideUsed) HIP_CHECK(hipEventRecord(evSide_, side_));
    sidePending_ = sideUsed;
}

// X = L11^-1 of the fronts of one level (see k_xinv_*), enqueued on `st`
void MfNumeric::enqueueInverses(int l, hipStream_t st)
{
    const XinvLevel& XL = xinvLevel_[l];
    TreeView tv{ frontOff_.p, idxPtr_.p, firstNode_.p, childPtr_.p, child_.p, invPtr_.p, inv_.p, idx_.p, dinvOff_.p };
    XinvView xv{ xinvOff_.p, xinvX_.p, xinvT_.p };
    hipLaunchKernelGGL(k_xinv_init, dim3(XL.init.cnt), dim3(256), 0, st, xinvDesc_.p + XL.init.off, tv, xv, dinv_.p);
    for (const auto& R : XL.rounds) {
        if (R.first.cnt) hipLaunchKernelGGL(k_xinv_gemm, dim3(R.first.cnt), dim3(256), 0, st, xinvDesc_.p + 2 * (size_t)R.first.off, tv, xv, fronts_.p);
        if (R.second.cnt) hipLaunchKernelGGL(k_xinv_gemm, dim3(R.second.cnt), dim3(256), 0, st, xinvDesc_.p + 2 * (size_t)R.second.off, tv, xv, fronts_.p);
    }
}

} // namespace ipcgpu
