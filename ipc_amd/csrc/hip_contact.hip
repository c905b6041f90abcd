// Contact kernels + HipContact host logic (gfx950).  Compiled with -ffp-contract=off (exact-comparison typing).
#include <climits>
#include "hip_contact.h"
#include "contact_device.h"
#include "stencil_hessian_device.h"
#include "orient3d_exact.h"
#include "hip_ipc.h"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <set>

namespace ipcgpu {

namespace {

using namespace cdev;
constexpr int BLOCK = 256;

struct ContactView {
    int nA, nP;
    const int* active; // int4 per entry
    const int* para;
    const int* paraEIEJ; // int2 per entry
    const int* SFE; // int2 per surface edge
    const double* x;
    const double* xRest;
    const unsigned char* need = nullptr; // owner-computes sharding: per node, does this rank own rows that the node's stencils add to?  null = every stencil
};

struct Stencil {
    int kind, n, node[4];
    double mult;
};
__device__ __forceinline__ Stencil decode(const int* c)
{
    Stencil s;
    s.mult = 1.0;
    s.node[2] = s.node[3] = 0;
    if (c[0] >= 0) {
        s.kind = K_EE;
        s.n = 4;
        for (int i = 0; i < 4; ++i) s.node[i] = c[i];
    }
    else {
        s.node[0] = -c[0] - 1;
        s.node[1] = c[1];
        if (c[2] < 0) {
            s.kind = K_PP;
            s.n = 2;
            s.mult = -c[3];
        }
        else if (c[3] < 0) {
            s.kind = K_PE;
            s.n = 3;
            s.node[2] = c[2];
            s.mult = -c[3];
        }
        else {
            s.kind = K_PT;
            s.n = 4;
            s.node[2] = c[2];
            s.node[3] = c[3];
        }
    }
    return s;
}
__device__ __forceinline__ void gatherX(const double* x, const int* node, int n, double (*X)[3])
{
    for (int k = 0; k < 4; ++k)
        for (int c = 0; c < 3; ++c) X[k][c] = (k < n) ? x[3 * (size_t)node[k] + c] : 0.0;
}
__device__ __forceinline__ double eps_x_of(const double* xr, int a0, int a1, int b0, int b1)
{
    double la = 0.0, lb = 0.0;
    for (int c = 0; c < 3; ++c) {
        const double da = xr[3 * (size_t)a0 + c] - xr[3 * (size_t)a1 + c];
        const double db = xr[3 * (size_t)b0 + c] - xr[3 * (size_t)b1 + c];
        la += da * da;
        lb += db * db;
    }
    return 1.0e-3 * la * lb; // MeshCollisionUtils.hpp:2969-2974
}
__device__ __forceinline__ void paraNodes(const ContactView& cv, int i, int* en)
{
    const int* c = cv.para + 4 * (size_t)i;
    if (c[3] >= 0) {
        for (int k = 0; k < 4; ++k) en[k] = c[k];
    }
    else {
        const int eI = cv.paraEIEJ[2 * (size_t)i], eJ = cv.paraEIEJ[2 * (size_t)i + 1];
        en[0] = cv.SFE[2 * (size_t)eI];
        en[1] = cv.SFE[2 * (size_t)eI + 1];
        en[2] = cv.SFE[2 * (size_t)eJ];
        en[3] = cv.SFE[2 * (size_t)eJ + 1];
    }
}
__device__ __forceinline__ double dist_only(int kind, const double (*X)[3])
{
    switch (kind) {
    case K_PP: return d_PP(X[0], X[1]);
    case K_PE: return d_PE(X[0], X[1], X[2]);
    case K_PT: return d_PT(X[0], X[1], X[2], X[3]);
    default: return d_EE(X[0], X[1], X[2], X[3]);
    }
}
__device__ __forceinline__ bool projected_dbc(int type, int projectDBC) { return type == 1 || (type == 2 && projectDBC); }

__device__ __forceinline__ double block_sum(double x, double* sm)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sm[wv] = x;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < BLOCK / 64; ++i) r += sm[i];
    return r;
}

// sum mult b(d) + sum e b(d)   (Optimizer.cpp:3252-3353), fixed-order reduction
__global__ __launch_bounds__(BLOCK) void k_contact_energy(ContactView cv, double dHat, double* __restrict__ partial)
{
    __shared__ double sm[BLOCK / 64];
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    double val = 0.0;
    if (i < cv.nA) {
        const Stencil s = decode(cv.active + 4 * (size_t)i);
        double X[4][3], b, gb, Hb;
        gatherX(cv.x, s.node, s.n, X);
        barrier(dist_only(s.kind, X), dHat, &b, &gb, &Hb);
        val = b * s.mult;
    }
    else if (i < cv.nA + cv.nP) {
        const int j = i - cv.nA;
        Stencil s = decode(cv.para + 4 * (size_t)j);
        double X[4][3], b, gb, Hb, e, eg, eH;
        gatherX(cv.x, s.node, s.n, X);
        barrier(dist_only(s.kind, X), dHat, &b, &gb, &Hb);
        int en[4];
        paraNodes(cv, j, en);
        double XE[4][3];
        gatherX(cv.x, en, 4, XE);
        mollifier(cross_sqnorm(XE[0], XE[1], XE[2], XE[3]), eps_x_of(cv.xRest, en[0], en[1], en[2], en[3]), &e, &eg, &eH);
        val = b * e;
    }
    const double r = block_sum(val, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
// pubWords > 0: the block also copies pubWords 32-bit words pubSrc -> pubDst (mapped host memory) once its own result is written -- the read-back of a batch
// of scalars that ends with this reduction needs no launch of its own
__global__ __launch_bounds__(BLOCK) void k_reduce_scaled(const double* __restrict__ partial, int n, double scale, double* __restrict__ out,
    const unsigned* pubSrc, unsigned* pubDst, int pubWords)
{
    __shared__ double sm[BLOCK / 64];
    double x = 0.0;
    for (int i = threadIdx.x; i < n; i += BLOCK) x += partial[i];
    const double r = block_sum(x, sm);
    if (threadIdx.x == 0) out[0] = scale * r;
    if (pubWords > 0) {
        __threadfence();
        __syncthreads();
        // (agent-scope loads: out[0] was written by this block a moment ago and must not come from a stale line of the vector cache)
        if ((int)threadIdx.x < pubWords) pubDst[threadIdx.x] = __hip_atomic_load(pubSrc + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- deterministic scatter ------------------------------------------------------------------------------------------------------------
// Barrier and friction terms add into nodes / CSR blocks that several stencils share.  Rounds 1-2 used hardware fp64 atomics: the last
// bits of g and a[] then depend on the order the atomics retire.  Now every stencil writes its contributions into its OWN slots of a
// scratch array together with a key (the node, or the CSR position of the 3x3 block); the slots of a key are brought into slot order
// (= stencil order) and summed front to back into the destination -- a fixed summation order, bit-reproducible.  (The atomic path stayed
// behind an environment switch for A/B timing until round 6: profiles/r03_contact_bench_atomic_scatter.json against
// r03_contact_bench_deterministic_scatter.json.)
constexpr unsigned KEY_NONE = 0xFFFFFFFFu;
// Deterministic scatter (round 6: a counting sort; rounds 3-5 sorted the keys with rocPRIM's radix sort, which is a merge sort of ~16 launches at these sizes):
// every contribution has a slot of its own (stencil index x slots per stencil), a key (the node, or the CSR position of the 3 x 3 block) and bumps the
// counter of its key's bucket; one scan lays the buckets out, k_det_fill drops the slot indices into them (counting the counters back down to zero), k_det_rank puts every
// bucket into ascending slot order, and one thread per bucket adds its contributions up in that order -- the order the stable sort produced, so the sums are bit for bit the same.
struct GradSink {
    double* contrib; // 3 doubles per slot
    unsigned* key; // node per slot (KEY_NONE: nothing to add)
    int* count; // contributions per node
};
__device__ __forceinline__ void sink_add3(const GradSink& k, size_t slot, int node, const double v[3])
{
    k.key[slot] = (unsigned)node;
    atomicAdd(k.count + node, 1);
    k.contrib[3 * slot] = v[0];
    k.contrib[3 * slot + 1] = v[1];
    k.contrib[3 * slot + 2] = v[2];
}
struct BlockSink {
    double* contrib; // 9 doubles per slot, entry (r, c) at r + 3 c
    unsigned* key; // CSR index of the block's first entry (KEY_NONE: nothing to add)
    int* rowNode; // row node of the block (the reducer derives row length and diagonal / off-diagonal from it)
    int* count; // contributions per block: bucket = key / 3 (the first entries of two blocks are at least three positions apart)
};
__device__ __forceinline__ void sink_key_block(const BlockSink& k, size_t slot, int p0, int rowNode)
{
    k.key[slot] = (unsigned)p0;
    k.rowNode[slot] = rowNode;
    atomicAdd(k.count + p0 / 3, 1);
}
// one thread per slot: its place in the bucket of its key
__global__ __launch_bounds__(BLOCK) void k_det_fill(int n, const unsigned* __restrict__ keys, int div, const int* __restrict__ start, int* __restrict__ count,
    int* __restrict__ seg)
{
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= n) return;
    const unsigned key = keys[t];
    if (key == KEY_NONE) return;
    const int b = (int)(key / (unsigned)div);
    seg[start[b] + atomicSub(count + b, 1) - 1] = t;
}
// one thread per bucket ENTRY: its rank among the slot indices of its bucket (a few tens of independent loads that hit the cache; one thread per bucket
// insertion-sorting in place was a chain of dependent global loads -- 0.12 ms for the fullest nodes), the bucket in ascending slot order into `sorted`
__global__ __launch_bounds__(BLOCK) void k_det_rank(int n, const unsigned* __restrict__ keys, int div, const int* __restrict__ start, int nKeys,
    const int* __restrict__ seg, int* __restrict__ sorted)
{
    const int u = blockIdx.x * BLOCK + threadIdx.x;
    if (u >= n || u >= start[nKeys]) return;
    const int slot = seg[u];
    const int b = (int)(keys[slot] / (unsigned)div);
    const int s0 = start[b], s1 = start[b + 1];
    int rank = 0;
    for (int w = s0; w < s1; ++w) rank += seg[w] < slot ? 1 : 0;
    sorted[s0 + rank] = slot;
}
// one thread per node: the node's contributions summed in slot order
__global__ __launch_bounds__(BLOCK) void k_bucket_sum3(int nKeys, const int* __restrict__ start, const int* __restrict__ seg, const double* __restrict__ contrib,
    double* __restrict__ grad)
{
    const int v = blockIdx.x * BLOCK + threadIdx.x;
    if (v >= nKeys) return;
    const int s0 = start[v], s1 = start[v + 1];
    if (s0 == s1) return;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    for (int u = s0; u < s1; ++u) {
        const double* q = contrib + 3 * (size_t)seg[u];
        t0 += q[0];
        t1 += q[1];
        t2 += q[2];
    }
    double* g = grad + 3 * (size_t)v;
    g[0] += t0;
    g[1] += t1;
    g[2] += t2;
}
// one thread per bucket of CSR positions (at most one block's first entry falls into it)
__global__ __launch_bounds__(BLOCK) void k_bucket_sum_blocks(int nKeys, const int* __restrict__ start, const int* __restrict__ seg, const unsigned* __restrict__ keys,
    const double* __restrict__ contrib, const int* __restrict__ rowNode, const int* __restrict__ ia, double* __restrict__ a)
{
    const int b = blockIdx.x * BLOCK + threadIdx.x;
    if (b >= nKeys) return;
    const int s0 = start[b], s1 = start[b + 1];
    if (s0 == s1) return;
    double S[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) S[k] = 0.0;
    for (int u = s0; u < s1; ++u) {
        const double* q = contrib + 9 * (size_t)seg[u];
#pragma unroll
        for (int k = 0; k < 9; ++k) S[k] += q[k];
    }
    const int vi = rowNode[seg[s0]], p0 = (int)keys[seg[s0]];
    const int base = ia[3 * vi], L = ia[3 * vi + 1] - base;
    if (p0 == base) { // the diagonal block: its upper triangle
        a[p0 + 0] += S[0];
        a[p0 + 1] += S[3];
        a[p0 + 2] += S[6];
        a[p0 + L + 0] += S[4];
        a[p0 + L + 1] += S[7];
        a[p0 + 2 * L - 1] += S[8];
    }
    else {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int rowOff = (r == 0) ? 0 : (r == 1 ? (L - 1) : (2 * L - 3));
#pragma unroll
            for (int c = 0; c < 3; ++c) a[p0 + rowOff + c] += S[r + 3 * c];
        }
    }
}

// grad += kappa (mult b' grad d)  resp.  kappa (b e' grad c + e b' grad d)   (SelfCollisionHandler.cpp:84-148, 2990-3036)
// slots: 8 per stencil -- 0..3 the nodes of the distance stencil (active list) resp. the four edge nodes (mollified list), 4..7 the
// nodes of the distance stencil of a mollified pair
// owner-computes sharding: a stencil is evaluated by every rank that owns rows of one of its nodes (stencils on a cut are evaluated more than once,
// like the elements on the rim of a patch); what it adds to the other nodes' rows is never read on this rank
__device__ __forceinline__ bool stencil_skipped(const unsigned char* need, const int* node, int n)
{
    if (!need) return false;
    for (int k = 0; k < n; ++k)
        if (need[node[k]]) return false;
    return true;
}

__global__ __launch_bounds__(BLOCK) void k_contact_gradient(ContactView cv, double dHat, double kappa, GradSink sink)
{
    // every one of a stencil's 8 slots gets its key here (KEY_NONE where there is nothing to add): no fill pass over the keys (round 6)
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < cv.nA) {
        const Stencil s = decode(cv.active + 4 * (size_t)i);
        const bool skip = stencil_skipped(cv.need, s.node, s.n);
        for (int k = skip ? 0 : s.n; k < 8; ++k) sink.key[8 * (size_t)i + k] = KEY_NONE;
        if (skip) return;
        double X[4][3], g[12], b, gb, Hb;
        gatherX(cv.x, s.node, s.n, X);
        const double d = stencil_distance(s.kind, X, g, nullptr);
        barrier(d, dHat, &b, &gb, &Hb);
        const double coef = kappa * s.mult * gb;
        for (int k = 0; k < s.n; ++k) {
            const double v[3] = { coef * g[3 * k], coef * g[3 * k + 1], coef * g[3 * k + 2] };
            sink_add3(sink, 8 * (size_t)i + k, s.node[k], v);
        }
    }
    else if (i < cv.nA + cv.nP) {
        const int j = i - cv.nA;
        const Stencil s = decode(cv.para + 4 * (size_t)j);
        int en[4];
        paraNodes(cv, j, en);
        const bool skip = stencil_skipped(cv.need, en, 4); // the edge pair's four nodes contain the stencil's
        for (int k = skip ? 0 : 4 + s.n; k < 8; ++k) sink.key[8 * (size_t)i + k] = KEY_NONE;
        if (skip) return;
        double X[4][3], g[12], b, gb, Hb;
        gatherX(cv.x, s.node, s.n, X);
        const double d = stencil_distance(s.kind, X, g, nullptr);
        barrier(d, dHat, &b, &gb, &Hb);
        double XE[4][3], cg[12], e, eg, eH;
        gatherX(cv.x, en, 4, XE);
        const double c = cross_sqnorm_derivs(XE, cg, nullptr);
        mollifier(c, eps_x_of(cv.xRest, en[0], en[1], en[2], en[3]), &e, &eg, &eH);
        for (int k = 0; k < 4; ++k) {
            const double v[3] = { kappa * b * eg * cg[3 * k], kappa * b * eg * cg[3 * k + 1], kappa * b * eg * cg[3 * k + 2] };
            sink_add3(sink, 8 * (size_t)i + k, en[k], v);
        }
        for (int k = 0; k < s.n; ++k) {
            const double v[3] = { kappa * e * gb * g[3 * k], kappa * e * gb * g[3 * k + 1], kappa * e * gb * g[3 * k + 2] };
            sink_add3(sink, 8 * (size_t)i + 4 + k, s.node[k], v);
        }
    }
}
// ---- the reference's per-constraint interface (round 5: what the compiled collision-handler adapter include/adapters/HipSelfCollisionHandler.hpp forwards to)
// SelfCollisionHandler::evaluateConstraints (SelfCollisionHandler.cpp:37-81): the squared distance of every given MMCVID tuple
__global__ __launch_bounds__(BLOCK) void k_evaluate_tuples(int n, const int* __restrict__ tuples, const double* __restrict__ x, double* __restrict__ val)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const Stencil s = decode(tuples + 4 * (size_t)i);
    double X[4][3];
    gatherX(x, s.node, s.n, X);
    val[i] = dist_only(s.kind, X);
}
// SelfCollisionHandler::leftMultiplyConstraintJacobianT (:84-148): out += coef * mult_i * input_i * grad d_i (mult = the multiplicity of a PP / PE tuple)
__global__ __launch_bounds__(BLOCK) void k_jt_tuples(int n, const int* __restrict__ tuples, const double* __restrict__ x, const double* __restrict__ input, double coef,
    double* __restrict__ out)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const Stencil s = decode(tuples + 4 * (size_t)i);
    double X[4][3], g[12];
    gatherX(x, s.node, s.n, X);
    (void)stencil_distance(s.kind, X, g, nullptr);
    const double w = coef * s.mult * input[i];
    for (int k = 0; k < s.n; ++k)
        for (int c = 0; c < 3; ++c) atomicAdd(&out[3 * (size_t)s.node[k] + c], w * g[3 * k + c]);
}
__global__ void k_zero_projected(int nV, const int* __restrict__ dbc, int projectDBC, double* __restrict__ grad)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < nV && projected_dbc(dbc[v], projectDBC)) grad[3 * (size_t)v] = grad[3 * (size_t)v + 1] = grad[3 * (size_t)v + 2] = 0.0; // Optimizer.cpp:3512-3516
}

struct CsrView {
    const int* ia;
    const int* ja;
};
// position of column 3*cn in row 3*rn of the upper CSR (cn > rn), -1 if absent
__device__ __forceinline__ int find_block(const CsrView& m, int rn, int cn)
{
    int lo = m.ia[3 * rn] + 3, hi = m.ia[3 * rn + 1]; // neighbour blocks start behind the 3 diagonal-block entries
    const int target = 3 * cn;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int c = m.ja[mid];
        if (c < target) lo = mid + 1;
        else hi = mid;
    }
    return (lo < m.ia[3 * rn + 1] && m.ja[lo] == target) ? lo : -1;
}
// does the CSR pattern hold a block for every node pair the barrier Hessian of the current sets will write?  (flag |= 1 if not)
__global__ void k_pattern_check(ContactView cv, CsrView m, int* __restrict__ flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int node[4], n = 0;
    if (i < cv.nA) {
        const Stencil s = decode(cv.active + 4 * (size_t)i);
        n = s.n;
        for (int k = 0; k < 4; ++k) node[k] = s.node[k];
    }
    else if (i < cv.nA + cv.nP) {
        paraNodes(cv, i - cv.nA, node);
        n = 4;
    }
    bool miss = false;
    for (int a = 0; a < n; ++a)
        for (int b = a + 1; b < n; ++b) {
            const int lo = min(node[a], node[b]), hi = max(node[a], node[b]);
            if (lo != hi && find_block(m, lo, hi) < 0) miss = true;
        }
    if (miss) atomicOr(flag, 1);
}

// ---- a += PSD-projected barrier Hessians (SelfCollisionHandler.cpp:418-561, 3039-3201) --------------------------------------------------------
// Round 6 (stencil_hessian_device.h): the stencil kind is a compile-time constant of the code a wave runs.  k_bin_stencils sorts the indices of
// the two lists into eight bins (active PP / PE / PT / EE, mollified with a PP / PE / PT / EE distance stencil); one workgroup = one wave of
// k_contact_hessian takes 64 stencils of ONE bin (the workgroup finds its bin from the bin counts, which never leave the device), forms the
// reduced block in the 9 x 9 frame, projects it (one instance of the Jacobi code for all bins: the rotations that only touch the zero rows of
// a 3 x 3 or 6 x 6 block are skipped by the whole wave) and writes the <= 10 node-pair blocks into the stencil's own slots of the
// deterministic scatter (HSLOTS per stencil, at HSLOTS * stencil index -- the order inside a bin does not matter).  Every slot of every
// stencil is written, with KEY_NONE where there is nothing to add (fewer nodes, a projected Dirichlet node, a stencil of another rank): no
// fill pass over the keys.  No scratch: every array index is a compile-time constant.
constexpr int HESS_W = 64; // one wave per workgroup
constexpr int HSLOTS = 10; // node pairs (k, l), k <= l, of a four-node stencil, in the order of pair_slot
constexpr int NBINS = 8;
struct HessBins {
    const int* count; // NBINS
    const int* perm; // NBINS x stride
    int stride;
};
__host__ __device__ constexpr int pair_slot(int k, int l) { return l * (l + 1) / 2 + k; } // k <= l < 4

__global__ __launch_bounds__(BLOCK) void k_bin_stencils(ContactView cv, int* __restrict__ count, int* __restrict__ perm, int stride)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    int bin = -1;
    if (i < cv.nA) bin = decode(cv.active + 4 * (size_t)i).kind;
    else if (i < cv.nA + cv.nP) bin = 4 + decode(cv.para + 4 * (size_t)(i - cv.nA)).kind;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int b = 0; b < NBINS; ++b) {
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(bin == b);
        if (!mask) continue;
        int base = 0;
        if (lane == __builtin_ctzll(mask)) base = atomicAdd(count + b, __builtin_popcountll(mask));
        base = __shfl(base, __builtin_ctzll(mask), 64);
        if (bin == b) perm[(size_t)b * stride + base + __builtin_popcountll(mask & ((1ull << lane) - 1))] = i;
    }
}

struct HessOut {
    CsrView m;
    BlockSink sink;
    int* err;
};
// one node-pair block of a projected stencil into its slot: rows from node K, columns from node L of the stencil (transposed when L's node
// comes first in the matrix: the upper CSR holds (min, max))
template <int NR, int K, int L>
__device__ __forceinline__ void emit_pair(const HessOut& o, const double* C, size_t slot0, const int* node, const bool* proj)
{
    const size_t slot = slot0 + pair_slot(K, L);
    if (proj[K] || proj[L]) {
        o.sink.key[slot] = KEY_NONE;
        return;
    }
    double A[9];
    sh::pair_block<NR, K, L, 9>(C, A);
    const int vk = node[K], vl = node[L];
    const bool swap = vk > vl;
    const int vi = swap ? vl : vk, vj = swap ? vk : vl;
    int p0 = o.m.ia[3 * vi];
    if (K != L) {
        p0 = find_block(o.m, vi, vj);
        if (p0 < 0) {
            atomicOr(o.err, 1); // the pattern lacks this contact pair: set_pattern must include the connectivity
            o.sink.key[slot] = KEY_NONE;
            return;
        }
    }
    sink_key_block(o.sink, slot, p0, vi);
    double* q = o.sink.contrib + 9 * slot;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) q[r + 3 * c] = swap ? A[c + 3 * r] : A[r + 3 * c];
}
template <int NN>
__device__ __forceinline__ void emit_stencil(const HessOut& o, const double* C, size_t slot0, const int* node, const int* __restrict__ dbc, int projectDBC)
{
    constexpr int NR = NN - 1;
    bool proj[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) proj[k] = k < NN ? projected_dbc(dbc[node[k]], projectDBC) : true;
    emit_pair<NR, 0, 0>(o, C, slot0, node, proj);
    emit_pair<NR, 0, 1>(o, C, slot0, node, proj);
    emit_pair<NR, 1, 1>(o, C, slot0, node, proj);
    if constexpr (NN >= 3) {
        emit_pair<NR, 0, 2>(o, C, slot0, node, proj);
        emit_pair<NR, 1, 2>(o, C, slot0, node, proj);
        emit_pair<NR, 2, 2>(o, C, slot0, node, proj);
    }
    if constexpr (NN >= 4) {
        emit_pair<NR, 0, 3>(o, C, slot0, node, proj);
        emit_pair<NR, 1, 3>(o, C, slot0, node, proj);
        emit_pair<NR, 2, 3>(o, C, slot0, node, proj);
        emit_pair<NR, 3, 3>(o, C, slot0, node, proj);
    }
#pragma unroll
    for (int p = NN * (NN + 1) / 2; p < HSLOTS; ++p) o.sink.key[slot0 + p] = KEY_NONE;
}
template <int KIND>
__device__ __forceinline__ void active_reduced(const ContactView& cv, const int* node, double mult, double dHat, double kappa, double (&C)[81])
{
    constexpr int NN = sh::NN_OF[KIND];
    double X[4][3];
#pragma unroll
    for (int k = 0; k < NN; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) X[k][c] = cv.x[3 * (size_t)node[k] + c];
    sh::active_block<KIND, 9>(X, dHat, kappa * mult, C);
}
template <int KIND>
__device__ __forceinline__ void para_reduced(const ContactView& cv, const int* node, const int* en, double dHat, double kappa, double (&C)[81])
{
    constexpr int NN = sh::NN_OF[KIND];
    double XE[4][3], sel[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int c = 0; c < 3; ++c) XE[q][c] = cv.x[3 * (size_t)en[q] + c];
#pragma unroll
        for (int k = 0; k < 4; ++k) sel[k][q] = (k < NN && en[q] == node[k]) ? 1.0 : 0.0; // the distance stencil's nodes are among the four (SelfCollisionHandler.cpp:3105-3160)
    }
    sh::para_block<KIND>(XE, sel, dHat, kappa, eps_x_of(cv.xRest, en[0], en[1], en[2], en[3]), C);
}

__global__ __launch_bounds__(HESS_W) void k_contact_hessian(ContactView cv, HessBins bins, CsrView m, const int* __restrict__ dbc, int projectDBC, double dHat,
    double kappa, BlockSink sink, int* __restrict__ err)
{
    // this workgroup's bin and its 64 entries (uniform: the counts are read through the scalar path)
    int bin = -1, first = 0, left = blockIdx.x, cnt = 0;
#pragma unroll
    for (int b = 0; b < NBINS; ++b) {
        const int c = bins.count[b], nb = (c + HESS_W - 1) / HESS_W;
        if (bin < 0) {
            if (left < nb) {
                bin = b;
                first = left * HESS_W;
                cnt = c;
            }
            else left -= nb;
        }
    }
    if (bin < 0) return;
    const int t = first + (int)threadIdx.x;
    const bool on = t < cnt;
    const bool isPara = bin >= 4;
    const int i = on ? bins.perm[(size_t)bin * bins.stride + t] : (isPara ? cv.nA : 0); // < nA: active list, else nA + index in the mollified list (a lane past the bin's end reads entry 0 of its list)
    const Stencil s = decode((isPara ? cv.para + 4 * (size_t)(i - cv.nA) : cv.active + 4 * (size_t)i));
    int en[4] = { s.node[0], s.node[1], s.node[2], s.node[3] }; // the nodes the block is scattered to
    if (isPara && on) paraNodes(cv, i - cv.nA, en);
    // owner-computes sharding: none of the stencil's nodes has rows on this rank (no early return: the wave votes in the Jacobi sweeps)
    const bool skip = !on || stencil_skipped(cv.need, en, isPara ? 4 : s.n);
    const HessOut o{ m, sink, err };
    const size_t slot0 = (size_t)HSLOTS * (size_t)i;
    double C[81];
    switch (bin) { // uniform
    case 0: active_reduced<K_PP>(cv, s.node, s.mult, dHat, kappa, C); break;
    case 1: active_reduced<K_PE>(cv, s.node, s.mult, dHat, kappa, C); break;
    case 2: active_reduced<K_PT>(cv, s.node, s.mult, dHat, kappa, C); break;
    case 3: active_reduced<K_EE>(cv, s.node, s.mult, dHat, kappa, C); break;
    case 4: para_reduced<K_PP>(cv, s.node, en, dHat, kappa, C); break;
    case 5: para_reduced<K_PE>(cv, s.node, en, dHat, kappa, C); break;
    case 6: para_reduced<K_PT>(cv, s.node, en, dHat, kappa, C); break;
    default: para_reduced<K_EE>(cv, s.node, en, dHat, kappa, C); break;
    }
    if (skip) { // a lane without a stencil of its own iterates on the zero matrix: converged before the first sweep
#pragma unroll
        for (int e = 0; e < 81; ++e) C[e] = 0.0;
    }
    int sweepsDone = sh::project_psd<9>(C);
    if (on) {
        if (skip) {
#pragma unroll
            for (int p = 0; p < HSLOTS; ++p) sink.key[slot0 + p] = KEY_NONE;
        }
        else if (bin == 0) emit_stencil<2>(o, C, slot0, en, dbc, projectDBC);
        else if (bin == 1) emit_stencil<3>(o, C, slot0, en, dbc, projectDBC);
        else emit_stencil<4>(o, C, slot0, en, dbc, projectDBC);
    }
    // total sweep count: a cheap health indicator (IPCGPU_DEBUG prints it); one atomic per wave
    for (int off = 32; off; off >>= 1) sweepsDone += __shfl_xor(sweepsDone, off);
    if ((threadIdx.x & 63) == 0 && sweepsDone) atomicAdd(err + 1, sweepsDone);
}

// ---- lagged friction (FrictionUtils.hpp:24-347, SelfCollisionHandler.cpp:2481-2988) --------------
// All four stencil kinds share one form: node weights wt_k and the 3 x 2 tangent basis B give T^T = [wt_k B^T]_k and the
// tangential sliding u = B^T sum_k wt_k (x_k - x_k^t).  One lane per lagged constraint.
struct FrictionView {
    int n;
    const int* set; // int4 MMCVID tuples (MMActiveSet_lastH)
    const double* lambda; // MMLambda_lastH
    const double* coord; // 2 per constraint (MMDistCoord)
    const double* basis; // 6 per constraint (MMTanBasis, column-major 3 x 2)
};
__device__ __forceinline__ void fr_weights(int kind, const double* co, double* wt)
{
    if (kind == K_PT) {
        wt[0] = 1.0;
        wt[1] = -1.0 + co[0] + co[1];
        wt[2] = -co[0];
        wt[3] = -co[1];
    }
    else if (kind == K_EE) {
        wt[0] = 1.0 - co[0];
        wt[1] = co[0];
        wt[2] = co[1] - 1.0;
        wt[3] = -co[1];
    }
    else if (kind == K_PE) {
        wt[0] = 1.0;
        wt[1] = co[0] - 1.0;
        wt[2] = -co[0];
        wt[3] = 0.0;
    }
    else {
        wt[0] = 1.0;
        wt[1] = -1.0;
        wt[2] = wt[3] = 0.0;
    }
}
__device__ __forceinline__ void fr_slide(const double* __restrict__ x, const double* __restrict__ xt, const Stencil& s, const double* wt,
    const double* B, double* u)
{
    double r[3] = { 0.0, 0.0, 0.0 };
    for (int k = 0; k < s.n; ++k)
        for (int c = 0; c < 3; ++c) r[c] += wt[k] * (x[3 * (size_t)s.node[k] + c] - xt[3 * (size_t)s.node[k] + c]);
    u[0] = B[0] * r[0] + B[1] * r[1] + B[2] * r[2];
    u[1] = B[3] * r[0] + B[4] * r[1] + B[5] * r[2];
}
__device__ __forceinline__ void fr_solve2(double a, double b, double d, double r0, double r1, double* xo)
{
    const double l10 = b / a, d1 = d - l10 * b;
    const double y1 = r1 - l10 * r0;
    xo[1] = y1 / d1;
    xo[0] = r0 / a - l10 * xo[1];
}
__device__ __forceinline__ void fr_normalize(double* a)
{
    const double l = sqrt(dot3(a, a));
    for (int i = 0; i < 3; ++i) a[i] /= l;
}
// multipliers, closest-point parameters and tangent bases at the current positions (Optimizer.cpp:1578-1598)
// obst / scaleSelf / scaleObst: a kinematic mesh obstacle carries its own friction coefficient (MeshCO::friction); the caller passes the
// larger coefficient to the friction terms and the lagged normal force of a stencil is scaled by the factor of its kind here
__global__ __launch_bounds__(BLOCK) void k_friction_lag(int n, const int* __restrict__ set, const double* __restrict__ x, double dHat, double kappa,
    const int* __restrict__ obst, double scaleSelf, double scaleObst, double* __restrict__ lambda, double* __restrict__ coord,
    double* __restrict__ basis)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const Stencil s = decode(set + 4 * (size_t)i);
    double X[4][3], b, gb, Hb;
    gatherX(x, s.node, s.n, X);
    const double d = dist_only(s.kind, X);
    barrier(d, dHat, &b, &gb, &Hb);
    double lam = gb * (-kappa * 2.0 * sqrt(d));
    if (set[4 * (size_t)i + 3] < -1) lam *= -set[4 * (size_t)i + 3];
    if (obst) {
        bool ob = false;
        for (int k = 0; k < s.n; ++k) ob = ob || obst[s.node[k]] != 0;
        lam *= ob ? scaleObst : scaleSelf;
    }
    lambda[i] = lam;
    double co[2] = { 0.0, 0.0 }, t0[3], t1[3], tmp[3];
    if (s.kind == K_EE) {
        double e20[3], e01[3], e23[3];
        sub3(X[0], X[2], e20);
        sub3(X[1], X[0], e01);
        sub3(X[3], X[2], e23);
        fr_solve2(dot3(e01, e01), -dot3(e23, e01), dot3(e23, e23), -dot3(e20, e01), dot3(e20, e23), co);
        for (int c = 0; c < 3; ++c) t0[c] = e01[c];
        cross3(e01, e23, tmp);
        cross3(tmp, e01, t1);
    }
    else if (s.kind == K_PT) {
        double e1[3], e2[3], w[3];
        sub3(X[2], X[1], e1);
        sub3(X[3], X[1], e2);
        sub3(X[0], X[1], w);
        fr_solve2(dot3(e1, e1), dot3(e1, e2), dot3(e2, e2), dot3(e1, w), dot3(e2, w), co);
        for (int c = 0; c < 3; ++c) t0[c] = e1[c];
        cross3(e1, e2, tmp);
        cross3(tmp, e1, t1);
    }
    else if (s.kind == K_PE) {
        double e12[3], w[3];
        sub3(X[2], X[1], e12);
        sub3(X[0], X[1], w);
        co[0] = dot3(w, e12) / dot3(e12, e12);
        for (int c = 0; c < 3; ++c) t0[c] = e12[c];
        cross3(e12, w, t1);
    }
    else {
        double v01[3], xC[3], yC[3];
        sub3(X[1], X[0], v01);
        const double ex[3] = { 1.0, 0.0, 0.0 }, ey[3] = { 0.0, 1.0, 0.0 };
        cross3(ex, v01, xC);
        cross3(ey, v01, yC);
        const bool px = dot3(xC, xC) > dot3(yC, yC);
        for (int c = 0; c < 3; ++c) t0[c] = px ? xC[c] : yC[c];
        cross3(v01, t0, t1);
    }
    fr_normalize(t0);
    fr_normalize(t1);
    coord[2 * (size_t)i] = co[0];
    coord[2 * (size_t)i + 1] = co[1];
    for (int c = 0; c < 3; ++c) {
        basis[6 * (size_t)i + c] = t0[c];
        basis[6 * (size_t)i + 3 + c] = t1[c];
    }
}
__global__ __launch_bounds__(BLOCK) void k_friction_energy(FrictionView fv, const double* __restrict__ x, const double* __restrict__ xt, double eps2,
    double* __restrict__ partial)
{
    __shared__ double sm[BLOCK / 64];
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    double val = 0.0;
    if (i < fv.n) {
        const Stencil s = decode(fv.set + 4 * (size_t)i);
        double wt[4], u[2];
        fr_weights(s.kind, fv.coord + 2 * (size_t)i, wt);
        fr_slide(x, xt, s, wt, fv.basis + 6 * (size_t)i, u);
        const double x2 = u[0] * u[0] + u[1] * u[1], eps = sqrt(eps2);
        val = fv.lambda[i] * ((x2 > eps2) ? sqrt(x2) : (x2 * (-sqrt(x2) / 3.0 + eps) / (eps * eps) + eps / 3.0)); // f0_SF_C1
    }
    const double r = block_sum(val, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
__global__ __launch_bounds__(BLOCK) void k_friction_gradient(FrictionView fv, const double* __restrict__ x, const double* __restrict__ xt, double eps2,
    double coef, GradSink sink)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= fv.n) return;
    const Stencil s = decode(fv.set + 4 * (size_t)i);
    const double* B = fv.basis + 6 * (size_t)i;
    double wt[4], u[2];
    fr_weights(s.kind, fv.coord + 2 * (size_t)i, wt);
    fr_slide(x, xt, s, wt, B, u);
    const double x2 = u[0] * u[0] + u[1] * u[1], eps = sqrt(eps2);
    const double sc = (x2 > eps2) ? 1.0 / sqrt(x2) : (-sqrt(x2) + 2.0 * eps) / (eps * eps); // f1 / |u|
    double t3[3];
    for (int c = 0; c < 3; ++c) t3[c] = B[c] * (u[0] * sc) + B[3 + c] * (u[1] * sc);
    for (int k = 0; k < s.n; ++k) {
        const double v[3] = { coef * fv.lambda[i] * wt[k] * t3[0], coef * fv.lambda[i] * wt[k] * t3[1], coef * fv.lambda[i] * wt[k] * t3[2] };
        sink_add3(sink, 8 * (size_t)i + k, s.node[k], v);
    }
}
// H = T^T (aI I + bU u u^T) T.  With orthonormal B, T T^T is a multiple of the identity, so the eigenvalues of H are those of
// the 2 x 2 core: aI and aI + bU |u|^2, both >= 0 in either regime (sliding: lambda/|u| and 0; sticking: lambda f1/|u| and
// lambda f2).  The reference's makePD (SelfCollisionHandler.cpp:2783, 2802 ...) is therefore the identity up to round-off
// and no eigen-solve is needed here; the oracle keeps it and the two agree to 1e-9.
__global__ __launch_bounds__(BLOCK) void k_friction_hessian(FrictionView fv, CsrView m, const double* __restrict__ x, const double* __restrict__ xt,
    const int* __restrict__ dbc, int projectDBC, double eps2, double coef, BlockSink sink, int* __restrict__ err)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= fv.n) return;
    const Stencil s = decode(fv.set + 4 * (size_t)i);
    const double* B = fv.basis + 6 * (size_t)i;
    double wt[4], u[2];
    fr_weights(s.kind, fv.coord + 2 * (size_t)i, wt);
    fr_slide(x, xt, s, wt, B, u);
    const double x2 = u[0] * u[0] + u[1] * u[1], xn = sqrt(x2), eps = sqrt(eps2);
    const double cl = coef * fv.lambda[i];
    double aI, bU;
    if (x2 > eps2) {
        aI = cl / xn;
        bU = -cl / (x2 * xn);
    }
    else {
        const double f1d = (-xn + 2.0 * eps) / (eps * eps), f2 = 2.0 * (eps - xn) / (eps * eps);
        aI = cl * f1d;
        bU = (f2 != f1d && x2 != 0.0) ? cl * (f2 - f1d) / x2 : 0.0;
    }
    double bu[3], BBt[9];
    for (int c = 0; c < 3; ++c) bu[c] = B[c] * u[0] + B[3 + c] * u[1];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) BBt[r + 3 * c] = aI * (B[r] * B[c] + B[3 + r] * B[3 + c]) + bU * bu[r] * bu[c];
    for (int k = 0; k < s.n; ++k) {
        const int vi = s.node[k];
        if (projected_dbc(dbc[vi], projectDBC)) continue;
        for (int l = 0; l < s.n; ++l) {
            const int vj = s.node[l];
            if (projected_dbc(dbc[vj], projectDBC) || vi > vj) continue;
            const double w = wt[k] * wt[l];
            // the weighted block into this stencil's slot (4 k + l) of the deterministic scatter
            if (vi == vj && l != k) continue;
            int p0 = m.ia[3 * vi];
            if (vi != vj) {
                p0 = find_block(m, vi, vj);
                if (p0 < 0) {
                    atomicOr(err, 1);
                    continue;
                }
            }
            const size_t slot = 16 * (size_t)i + 4 * k + l;
            sink_key_block(sink, slot, p0, vi);
            for (int q = 0; q < 9; ++q) sink.contrib[9 * slot + q] = w * BBt[q];
        }
    }
}

// ---- broad phase: uniform grid by counting sort ---------------------------------------------------------------
struct Grid {
    double lo[3], h;
    int dim[3];
};
__device__ __forceinline__ int cell_of(const Grid& g, double x, int c)
{
    return min(g.dim[c] - 1, max(0, (int)floor((x - g.lo[c]) / g.h)));
}
// Pair flags of a node as the broad / narrow phase kernels read them: bit 0 Dirichlet node, bit 1 node of a kinematic
// obstacle (the reference's MeshCO riding along as a surface-only component), bit 2 "only pairs that involve an obstacle"
// (a scene with `meshCO` and `selfCollisionOff`; set on every node so that either operand tells).
__device__ __forceinline__ bool pair_filtered(int fa, int fb) { return (fa & 4) && !((fa | fb) & 2); }
__global__ void k_pair_flags(int nV, const int* __restrict__ dbc, const int* __restrict__ obst, int obstacleOnly, int* __restrict__ flags)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < nV) flags[v] = (dbc[v] != 0 ? 1 : 0) | ((obst && obst[v]) ? 2 : 0) | (obstacleOnly ? 4 : 0);
}

__global__ __launch_bounds__(BLOCK) void k_bbox_partial(int nV, const double* __restrict__ x, double* __restrict__ partial)
{
    __shared__ double sm[6][BLOCK / 64];
    const int v = blockIdx.x * BLOCK + threadIdx.x;
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    if (v < nV)
        for (int c = 0; c < 3; ++c) lo[c] = hi[c] = x[3 * (size_t)v + c];
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[c] = fmin(lo[c], __shfl_down(lo[c], off, 64));
            hi[c] = fmax(hi[c], __shfl_down(hi[c], off, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            sm[c][threadIdx.x >> 6] = lo[c];
            sm[3 + c][threadIdx.x >> 6] = hi[c];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        double r = sm[threadIdx.x][0];
        for (int i = 1; i < BLOCK / 64; ++i) r = (threadIdx.x < 3) ? fmin(r, sm[threadIdx.x][i]) : fmax(r, sm[threadIdx.x][i]);
        partial[6 * (size_t)blockIdx.x + threadIdx.x] = r;
    }
}
// the box of the current positions; *stale = 1 when it leaves the grid `g` the host laid over the previous build's box (the kernels of the build then return at
// once and the host repeats the build on a fresh grid: with everything clamped into the border cells one cell would hold the whole surface)
__global__ __launch_bounds__(BLOCK) void k_bbox_final(int nb, const double* __restrict__ partial, double* __restrict__ box6, Grid g, int check, int* __restrict__ stale)
{
    __shared__ double sm[6][BLOCK / 64];
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    for (int b = threadIdx.x; b < nb; b += BLOCK)
        for (int c = 0; c < 3; ++c) {
            lo[c] = fmin(lo[c], partial[6 * (size_t)b + c]);
            hi[c] = fmax(hi[c], partial[6 * (size_t)b + 3 + c]);
        }
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[c] = fmin(lo[c], __shfl_down(lo[c], off, 64));
            hi[c] = fmax(hi[c], __shfl_down(hi[c], off, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            sm[c][threadIdx.x >> 6] = lo[c];
            sm[3 + c][threadIdx.x >> 6] = hi[c];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        double r = sm[threadIdx.x][0];
        for (int i = 1; i < BLOCK / 64; ++i) r = (threadIdx.x < 3) ? fmin(r, sm[threadIdx.x][i]) : fmax(r, sm[threadIdx.x][i]);
        box6[threadIdx.x] = r;
        const int c = threadIdx.x % 3;
        if (check && (threadIdx.x < 3 ? r < g.lo[c] : r > g.lo[c] + g.h * g.dim[c])) atomicOr(stale, 1);
    }
}
// Candidate walks (narrow phase, CCD sweeps, intersection test).  A surface has ~1e5 primitives and a primitive ~1e2 candidates behind a chain
// of dependent loads (cell range -> item -> its nodes -> their positions): one lane per primitive leaves the machine two waves per SIMD, each
// lane serialising its chains.  What the walks cost in round 2 (k_narrow_ee 0.57 ms, k_ref_sweep_edge 0.83 ms, k_ref_sweep_vertex 0.56 ms for
// 1.2e5 edges / 4e4 vertices) and what measurements on the contact benchmark said about it, in the order tried:
//   * COOP lanes share a primitive and stride over the ITEMS of each cell (eight times the waves in flight): narrow phase and intersection
//     test 2-3x faster, the sweeps unchanged.  Splitting the CELLS of a primitive's box over the lanes instead was 2.5x slower;
//   * 32-byte records in cell order instead of indices (below): no gather chain before the box test -- by itself no change either;
//   * what the sweeps were actually waiting for: a global counter bumped once per queried pair.  Same-address atomics retire at ~2.5 ns each
//     whatever the number of lanes: 3e5 of them ARE the 0.8 ms.  Counters now live in registers and are added once per wave at the end
//     (wave_count_add), found pairs are collected per workgroup in LDS (WgList), the limiting pair comes out of a hit list instead of a second
//     run of the sweep: k_ref_sweep_edge 2 x 0.83 -> 0.44 ms, k_ref_sweep_vertex 2 x 0.56 -> 0.11 ms, k_narrow_ee 0.27 ms, k_narrow_pt 0.036 ms.
//   Listing the candidates first and querying them one lane per pair in a second kernel (balanced queries) was tried as well: the list costs one
//   same-address atomic per group of lanes that reach the append together -- in a divergent walk almost one per pair -- and ran 5x slower.
constexpr int COOP = 8;
constexpr int SWEEP_COOP = 8;
// A slot of a device-side list for every lane that is active here: ONE atomic per wave (the lanes of a candidate walk reach this point a few
// at a time; a counter bumped once per candidate by 3e5 lanes serialises in the L2 atomic unit).
__device__ __forceinline__ int wave_slot(int* counter)
{
    const unsigned long long m = __ballot(1);
    const int lane = (int)__lane_id();
    int base = 0;
    if (lane == __ffsll((long long)m) - 1) base = atomicAdd(counter, __popcll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    return base + __popcll(m & ((1ull << lane) - 1ull));
}
// What a cell list holds.  With bare primitive indices, every candidate of a walk costs a chain of gathers -- index -> its nodes -> their positions
// (or node -> surface vertex -> voxel box): 3-5 scattered 32-64 B sectors per candidate, ~1e7 candidates per walk, and the walks sat at the L2's
// gather rate whatever the number of lanes.  The lists therefore hold 32-byte RECORDS in cell order -- the primitive and the box the insertion
// computed anyway -- so that a walk streams its candidates and touches a primitive's nodes only when the boxes overlap:
//   uniform grid (narrow phase, intersection test):  (index, float lo[3], float hi[3], 0), the inflated box rounded OUTWARDS to float -- a
//       conservative first test; the exact double test on the positions follows for the survivors, unchanged
//   reference voxel grid (CCD sweeps):               (index, int lo[3], int hi[3], 0), the voxel box itself -- the test on it is the exact one
constexpr int REC = 8; // ints per record
__device__ __forceinline__ float f_down(double v)
{
    const float f = (float)v;
    return ((double)f > v) ? nextafterf(f, -INFINITY) : f;
}
__device__ __forceinline__ float f_up(double v)
{
    const float f = (float)v;
    return ((double)f < v) ? nextafterf(f, INFINITY) : f;
}
struct BoxRec {
    int id;
    float lo[3], hi[3];
};
__device__ __forceinline__ BoxRec load_box_rec(const int* __restrict__ recs, int k)
{
    const int4 r0 = reinterpret_cast<const int4*>(recs)[2 * (size_t)k], r1 = reinterpret_cast<const int4*>(recs)[2 * (size_t)k + 1];
    return BoxRec{ r0.x, { __int_as_float(r0.y), __int_as_float(r0.z), __int_as_float(r0.w) }, { __int_as_float(r1.x), __int_as_float(r1.y), __int_as_float(r1.z) } };
}
struct VoxRec {
    int id;
    int b[6];
};
__device__ __forceinline__ VoxRec load_vox_rec(const int* __restrict__ recs, int k)
{
    const int4 r0 = reinterpret_cast<const int4*>(recs)[2 * (size_t)k], r1 = reinterpret_cast<const int4*>(recs)[2 * (size_t)k + 1];
    return VoxRec{ r0.x, { r0.y, r0.z, r0.w, r1.x, r1.y, r1.z } };
}
// mode 0: count, mode 1: fill (records, see above).  Primitive = triangle (isTri) or surface edge, bbox inflated by `infl`
__global__ __launch_bounds__(BLOCK) void k_grid_insert(int nPrim, int isTri, const int* __restrict__ prim, const double* __restrict__ x, Grid g,
    double infl, int mode, int* __restrict__ cellCount, const int* __restrict__ cellStart, int* __restrict__ cellItems)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nPrim) return;
    const int nv = isTri ? 3 : 2;
    double bl[3] = { 1e300, 1e300, 1e300 }, bh[3] = { -1e300, -1e300, -1e300 };
    for (int k = 0; k < nv; ++k) {
        const int v = prim[nv * (size_t)i + k];
        for (int c = 0; c < 3; ++c) {
            const double xv = x[3 * (size_t)v + c];
            bl[c] = fmin(bl[c], xv - infl);
            bh[c] = fmax(bh[c], xv + infl);
        }
    }
    int a[3], b[3];
    for (int c = 0; c < 3; ++c) {
        a[c] = cell_of(g, bl[c], c);
        b[c] = cell_of(g, bh[c], c);
    }
    for (int z = a[2]; z <= b[2]; ++z)
        for (int y = a[1]; y <= b[1]; ++y)
            for (int xx = a[0]; xx <= b[0]; ++xx) {
                const int cell = xx + g.dim[0] * (y + g.dim[1] * z);
                const int slot = atomicAdd(&cellCount[cell], 1);
                if (mode == 1) {
                    int4* r = reinterpret_cast<int4*>(cellItems) + 2 * (size_t)(cellStart[cell] + slot);
                    r[0] = make_int4(i, __float_as_int(f_down(bl[0])), __float_as_int(f_down(bl[1])), __float_as_int(f_down(bl[2])));
                    r[1] = make_int4(__float_as_int(f_up(bh[0])), __float_as_int(f_up(bh[1])), __float_as_int(f_up(bh[2])), 0);
                }
            }
}

// Triangles AND edges of the surface in one pass over one cell array of 2 nCells + 1 counters (round 6: the two grids of the narrow phase used to be built one
// after the other -- fill, count, scan, read-back, fill, insert for each): thread i < nTri takes triangle i into cells [0, nCells), the others edge i - nTri
// into [nCells, 2 nCells).  mode 0 counts; mode 1 fills and takes its slots by counting the cells' counters back DOWN to zero -- no second fill pass, and the
// array is zero again when the pass is over.
__global__ __launch_bounds__(BLOCK) void k_grid_insert_both(int nTri, const int* __restrict__ tri, int nEdge, const int* __restrict__ edge, const double* __restrict__ x,
    Grid g, int nCells, double infl, int mode, int capItems, const int* __restrict__ stale, int* __restrict__ cellCount, const int* __restrict__ cellStart,
    int* __restrict__ cellItems)
{
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= nTri + nEdge || *stale) return;
    const bool isTri = t < nTri;
    const int i = isTri ? t : t - nTri, nv = isTri ? 3 : 2, base = isTri ? 0 : nCells;
    const int* prim = isTri ? tri : edge;
    double bl[3] = { 1e300, 1e300, 1e300 }, bh[3] = { -1e300, -1e300, -1e300 };
    for (int k = 0; k < nv; ++k) {
        const int v = prim[nv * (size_t)i + k];
        for (int c = 0; c < 3; ++c) {
            const double xv = x[3 * (size_t)v + c];
            bl[c] = fmin(bl[c], xv - infl);
            bh[c] = fmax(bh[c], xv + infl);
        }
    }
    int a[3], b[3];
    for (int c = 0; c < 3; ++c) {
        a[c] = cell_of(g, bl[c], c);
        b[c] = cell_of(g, bh[c], c);
    }
    for (int z = a[2]; z <= b[2]; ++z)
        for (int y = a[1]; y <= b[1]; ++y)
            for (int xx = a[0]; xx <= b[0]; ++xx) {
                const int cell = base + xx + g.dim[0] * (y + g.dim[1] * z);
                if (mode == 0) atomicAdd(&cellCount[cell], 1);
                else {
                    const int slot = atomicSub(&cellCount[cell], 1) - 1;
                    if (cellStart[cell] + slot >= capItems) continue; // (the host sees the total, grows the list and repeats the build)
                    int4* r = reinterpret_cast<int4*>(cellItems) + 2 * (size_t)(cellStart[cell] + slot);
                    r[0] = make_int4(i, __float_as_int(f_down(bl[0])), __float_as_int(f_down(bl[1])), __float_as_int(f_down(bl[2])));
                    r[1] = make_int4(__float_as_int(f_up(bh[0])), __float_as_int(f_up(bh[1])), __float_as_int(f_up(bh[2])), 0);
                }
            }
}

// Output of the narrow phase: records of 6 ints = MMCVID (4) + (svI | eI, sfI | eJ), appended to a global list.  A workgroup collects its
// records in LDS and reserves its range of the list with ONE atomic at the end (7.8e4 records through one global counter were 0.2 ms of
// serialised same-address atomics); a workgroup that finds more than WG_RECS falls back to the global counter for the excess.
constexpr int WG_RECS = 512;
struct WgList {
    int* count; // LDS
    int* recs; // LDS, 6 * WG_RECS
    int* bucket; // global: records per first primitive (surface vertex | edge) -- the counting sort that orders the list afterwards starts here (k_bucket_fill)
};
__device__ __forceinline__ void wg_list_put(const WgList& w, const int* id, int i, int j, int cap, int* __restrict__ out, int* __restrict__ counter)
{
    atomicAdd(w.bucket + i, 1);
    int slot = atomicAdd(w.count, 1);
    int* o;
    if (slot < WG_RECS) o = w.recs + 6 * slot;
    else {
        slot = atomicAdd(counter, 1);
        if (slot >= cap) return;
        o = out + 6 * (size_t)slot;
    }
    o[0] = id[0]; o[1] = id[1]; o[2] = id[2]; o[3] = id[3];
    o[4] = i; o[5] = j;
}
__device__ __forceinline__ void wg_list_flush(const WgList& w, int* sBase, int cap, int* __restrict__ out, int* __restrict__ counter) // all threads
{
    __syncthreads();
    const int n = min(*w.count, WG_RECS);
    if (threadIdx.x == 0) *sBase = n ? atomicAdd(counter, n) : 0;
    __syncthreads();
    const int base = *sBase;
    for (int e = threadIdx.x; e < 6 * n; e += BLOCK) {
        const int slot = base + e / 6;
        if (slot < cap) out[6 * (size_t)slot + e % 6] = w.recs[e];
    }
}
__global__ __launch_bounds__(BLOCK) void k_narrow_pt(int nSVI, const int* __restrict__ SVI, const int* __restrict__ SF, const double* __restrict__ x,
    const int* __restrict__ dbc, Grid g, const int* __restrict__ cellStart, const int* __restrict__ cellItems, double dHat, int cap,
    int* __restrict__ out, int* __restrict__ counter, int capItems, const int* __restrict__ stale, int* __restrict__ bucket)
{
    __shared__ int sCount, sBase, sRecs[6 * WG_RECS];
    const WgList wl{ &sCount, sRecs, bucket };
    if (*stale) return; // (uniform: see k_bbox_final)
    if (threadIdx.x == 0) sCount = 0;
    __syncthreads();
    const int gi = blockIdx.x * BLOCK + threadIdx.x;
    const bool valid = gi / COOP < nSVI; // lanes past the end walk an empty range: every thread reaches the flush
    const int i = valid ? gi / COOP : 0, sub = gi % COOP; // COOP lanes share a vertex and stride over the cell's triangles (see COOP)
    const int vI = SVI[i];
    const double p[3] = { x[3 * (size_t)vI], x[3 * (size_t)vI + 1], x[3 * (size_t)vI + 2] };
    const int cell = cell_of(g, p[0], 0) + g.dim[0] * (cell_of(g, p[1], 1) + g.dim[1] * cell_of(g, p[2], 2));
    const int fI = dbc[vI]; // dbc: pair flags (bit 0 Dirichlet, bit 1 obstacle node, bit 2 obstacle-only filter on)
    const bool vDbc = (fI & 1) != 0;
    const int kEnd = valid ? min(cellStart[cell + 1], capItems) : 0; // (capItems: a list the fill pass had to truncate -- the host repeats the build)
    for (int k = cellStart[cell] + sub; k < kEnd; k += COOP) {
        const BoxRec rec = load_box_rec(cellItems, k);
        // a point closer than sqrt(dHat) to the triangle lies inside the triangle's box inflated by that much
        if (p[0] < rec.lo[0] || p[0] > rec.hi[0] || p[1] < rec.lo[1] || p[1] > rec.hi[1] || p[2] < rec.lo[2] || p[2] > rec.hi[2]) continue;
        const int f = rec.id;
        const int t0 = SF[3 * (size_t)f], t1 = SF[3 * (size_t)f + 1], t2 = SF[3 * (size_t)f + 2];
        // flags and positions requested together, filters without short-circuits (see narrow_ee_queued)
        const int f0 = dbc[t0], f1 = dbc[t1], f2 = dbc[t2];
        const double a[3] = { x[3 * (size_t)t0], x[3 * (size_t)t0 + 1], x[3 * (size_t)t0 + 2] };
        const double b[3] = { x[3 * (size_t)t1], x[3 * (size_t)t1 + 1], x[3 * (size_t)t1 + 2] };
        const double c[3] = { x[3 * (size_t)t2], x[3 * (size_t)t2 + 1], x[3 * (size_t)t2 + 2] };
        if ((vI == t0) | (vI == t1) | (vI == t2)) continue;
        if ((vDbc & ((f0 & f1 & f2 & 1) != 0)) | pair_filtered(fI, f0)) continue; // SelfCollisionHandler.cpp:2184-2187; the obstacle-only filter
        double d;
        int id[4] = { -vI - 1, -1, -1, -1 };
        switch (dType_PT(p, a, b, c)) {
        case 0: d = d_PP(p, a); id[1] = t0; break;
        case 1: d = d_PP(p, b); id[1] = t1; break;
        case 2: d = d_PP(p, c); id[1] = t2; break;
        case 3: d = d_PE(p, a, b); id[1] = t0; id[2] = t1; break;
        case 4: d = d_PE(p, b, c); id[1] = t1; id[2] = t2; break;
        case 5: d = d_PE(p, c, a); id[1] = t2; id[2] = t0; break;
        default: d = d_PT(p, a, b, c); id[1] = t0; id[2] = t1; id[3] = t2; break;
        }
        if (d < dHat) wg_list_put(wl, id, i, f, cap, out, counter);
    }
    wg_list_flush(wl, &sBase, cap, out, counter);
}

// typing, distance and the MMCVID tuple of one edge pair that passed the box tests (SelfCollisionHandler.cpp:2270-2400)
__device__ __forceinline__ void narrow_ee_pair(int eI, int eJ, int a0, int a1, int b0, int b1, const double* pa0, const double* pa1, const double* pb0,
    const double* pb1, const double* __restrict__ xRest, int nE, double dHat, const WgList& wl, int cap, int* __restrict__ out, int* __restrict__ counter)
{
    const int dt = dType_EE(pa0, pa1, pb0, pb1);
    const int add_e = (cross_sqnorm(pa0, pa1, pb0, pb1) < eps_x_of(xRest, a0, a1, b0, b1)) ? -eJ - 2 : -1;
    double d;
    int id[4];
    switch (dt) {
    case 0: d = d_PP(pa0, pb0); id[0] = -a0 - 1; id[1] = b0; id[2] = -1; id[3] = add_e; break;
    case 1: d = d_PP(pa0, pb1); id[0] = -a0 - 1; id[1] = b1; id[2] = -1; id[3] = add_e; break;
    case 2: d = d_PE(pa0, pb0, pb1); id[0] = -a0 - 1; id[1] = b0; id[2] = b1; id[3] = add_e; break;
    case 3: d = d_PP(pa1, pb0); id[0] = -a1 - 1; id[1] = b0; id[2] = -1; id[3] = add_e; break;
    case 4: d = d_PP(pa1, pb1); id[0] = -a1 - 1; id[1] = b1; id[2] = -1; id[3] = add_e; break;
    case 5: d = d_PE(pa1, pb0, pb1); id[0] = -a1 - 1; id[1] = b0; id[2] = b1; id[3] = add_e; break;
    case 6: d = d_PE(pb0, pa0, pa1); id[0] = -b0 - 1; id[1] = a0; id[2] = a1; id[3] = add_e; break;
    case 7: d = d_PE(pb1, pa0, pa1); id[0] = -b1 - 1; id[1] = a0; id[2] = a1; id[3] = add_e; break;
    default:
        d = d_EE(pa0, pa1, pb0, pb1);
        id[0] = a0; id[1] = a1; id[2] = b0;
        id[3] = (add_e <= -2) ? (-b1 - nE - 2) : b1;
        break;
    }
    if (d < dHat) wg_list_put(wl, id, eI, eJ, cap, out, counter);
}
// The edge-edge narrow phase CELL by cell (round 6).  The per-edge walk of rounds 2-5 (the shape k_narrow_pt still has) visited, for every edge, every record of every cell its inflated box touches (~300 records
// of 32 B per edge, most of them the same neighbours met again in the next cell and once more from the other edge's side), and whenever one of the eight
// lanes that share an edge finds a pair whose boxes overlap, it runs the typing + distance of that pair -- a few hundred fp64 instructions -- while the
// other lanes of the wave wait: 0.27 ms for 1.2e5 edges, the largest contact kernel for three rounds.  Here:
//   * one WAVE takes one cell.  Lane b holds record b of the cell with its edge's nodes and positions (loaded once); the wave walks the cell's records a
//     TOGETHER -- a uniform index, so record, nodes and positions of edge a arrive through the scalar cache, once per wave -- and lane b tests the pair
//     (a, b) when a's edge has the smaller index: outward-rounded float boxes, then the exact inflated boxes and the rule that a pair belongs to the one
//     cell that holds the low corner of their overlap (exactly as the per-edge walk had it: the SAME pairs survive);
//   * survivors are not typed where they are found: they go into a queue of the wave in LDS, and whenever 64 have gathered ALL lanes take one each --
//     typing and distance run on full waves.
// The same records come out (in another order; they are sorted by primitive pair afterwards).
constexpr int EE_QUEUE = 128;
__device__ __forceinline__ void narrow_ee_queued(int eI, int eJ, const int* __restrict__ SFE, const double* __restrict__ x, const double* __restrict__ xRest,
    const int* __restrict__ dbc, int nE, double dHat, const WgList& wl, int cap, int* __restrict__ out, int* __restrict__ counter)
{
    const int a0 = SFE[2 * (size_t)eI], a1 = SFE[2 * (size_t)eI + 1], b0 = SFE[2 * (size_t)eJ], b1 = SFE[2 * (size_t)eJ + 1];
    // flags and positions of the four nodes are requested together and the filters evaluated without short-circuits: as `dbc[a0] && dbc[a1] && ...` each
    // flag was a memory round trip of its own, in series, in front of the positions
    const int fa0 = dbc[a0], fa1 = dbc[a1], fb0 = dbc[b0], fb1 = dbc[b1];
    const double pa0[3] = { x[3 * (size_t)a0], x[3 * (size_t)a0 + 1], x[3 * (size_t)a0 + 2] };
    const double pa1[3] = { x[3 * (size_t)a1], x[3 * (size_t)a1 + 1], x[3 * (size_t)a1 + 2] };
    const double pb0[3] = { x[3 * (size_t)b0], x[3 * (size_t)b0 + 1], x[3 * (size_t)b0 + 2] };
    const double pb1[3] = { x[3 * (size_t)b1], x[3 * (size_t)b1 + 1], x[3 * (size_t)b1 + 2] };
    if (((fa0 & fa1 & fb0 & fb1 & 1) != 0) | pair_filtered(fa0, fb0)) return; // SelfCollisionHandler.cpp:2294-2297; the obstacle-only filter
    narrow_ee_pair(eI, eJ, a0, a1, b0, b1, pa0, pa1, pb0, pb1, xRest, nE, dHat, wl, cap, out, counter);
}
struct EeTileRec { // one edge of a cell's list as the pair loop reads it from LDS: everything the box tests need, fetched ONCE per cell by the lane that owns the record
    double lo[3], hi[3]; // the inflated box of the edge, exactly as the per-edge walk forms it
    int id, n0, n1;
    int c[3]; // cell of the box's low corner: cell_of is monotone, so the cell of the low corner of an overlap is the larger of the two (no division per pair)
};
__global__ __launch_bounds__(BLOCK) void k_narrow_ee_cells(int nE, const int* __restrict__ SFE, const double* __restrict__ x, const double* __restrict__ xRest,
    const int* __restrict__ dbc, Grid g, int nCells, const int* __restrict__ cellStart, const int* __restrict__ cellItems, double dHat, double infl, int cap,
    int* __restrict__ out, int* __restrict__ counter, int capItems, const int* __restrict__ stale, int* __restrict__ bucket)
{
    __shared__ int sCount, sBase, sRecs[6 * WG_RECS];
    __shared__ int2 sQueue[BLOCK / 64][EE_QUEUE];
    __shared__ EeTileRec sTile[BLOCK / 64][64];
    const WgList wl{ &sCount, sRecs, bucket };
    if (*stale) return; // (uniform: see k_bbox_final)
    if (threadIdx.x == 0) sCount = 0;
    __syncthreads();
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int2* q = sQueue[wv];
    EeTileRec* tile = sTile[wv];
    int qn = 0; // entries in the wave's queue (uniform)
    const int nWaves = gridDim.x * (BLOCK / 64);
    // a wave takes every nWaves-th cell: its queue fills across cells, so the typing below runs on full waves whatever a single cell yields
    for (int cell = blockIdx.x * (BLOCK / 64) + wv; cell < nCells; cell += nWaves) {
        const int kBeg = min(cellStart[cell], capItems), kEnd = min(cellStart[cell + 1], capItems); // (capItems: a truncated list -- the host repeats the build)
        if (kEnd - kBeg < 2) continue;
        const int cx = cell % g.dim[0], cy = (cell / g.dim[0]) % g.dim[1], cz = cell / (g.dim[0] * g.dim[1]);
        for (int kb0 = kBeg; kb0 < kEnd; kb0 += 64) {
            // this lane's record of the cell (registers): lanes past the end hold a copy of the first one and never pair
            const bool vb = kb0 + lane < kEnd;
            EeTileRec rb;
            {
                const int e = cellItems[(size_t)REC * (size_t)(vb ? kb0 + lane : kBeg)];
                rb.id = e;
                rb.n0 = SFE[2 * (size_t)e];
                rb.n1 = SFE[2 * (size_t)e + 1];
                for (int c = 0; c < 3; ++c) {
                    const double p0 = x[3 * (size_t)rb.n0 + c], p1 = x[3 * (size_t)rb.n1 + c];
                    rb.lo[c] = fmin(p0, p1) - infl;
                    rb.hi[c] = fmax(p0, p1) + infl;
                    rb.c[c] = cell_of(g, rb.lo[c], c);
                }
            }
            for (int ka0 = kBeg; ka0 <= kb0; ka0 += 64) { // tiles of the cell's list in LDS; a pair (a, b) with id_a < id_b is met once: a's tile <= b's chunk or the reverse
                const int nA = min(64, kEnd - ka0);
                __builtin_amdgcn_wave_barrier();
                if (ka0 == kb0) tile[lane] = rb; // the same 64 records: no second fetch
                else if (lane < nA) {
                    EeTileRec ra;
                    const int e = cellItems[(size_t)REC * (size_t)(ka0 + lane)];
                    ra.id = e;
                    ra.n0 = SFE[2 * (size_t)e];
                    ra.n1 = SFE[2 * (size_t)e + 1];
                    for (int c = 0; c < 3; ++c) {
                        const double p0 = x[3 * (size_t)ra.n0 + c], p1 = x[3 * (size_t)ra.n1 + c];
                        ra.lo[c] = fmin(p0, p1) - infl;
                        ra.hi[c] = fmax(p0, p1) + infl;
                        ra.c[c] = cell_of(g, ra.lo[c], c);
                    }
                    tile[lane] = ra;
                }
                __builtin_amdgcn_wave_barrier();
                for (int ka = 0; ka < nA; ++ka) {
                    const EeTileRec ra = tile[ka]; // uniform address: a broadcast read; by VALUE and tested without short-circuits below -- as a reference behind
                                                   // && chains every field was its own LDS round trip (fourteen in a row per pair, ~1.5 k cycles: 0.16 ms of the kernel)
                    // inflated boxes must overlap, and the pair is handled only in the cell holding the low corner of the overlap -- in either order of the two
                    // (different tiles meet once, with a's tile first; inside one tile both orders of (ka, lane) come by and the smaller edge index decides)
                    bool ok = vb & (ka0 == kb0 ? ra.id < rb.id : ra.id != rb.id);
#pragma unroll
                    for (int c = 0; c < 3; ++c) ok &= !(ra.lo[c] > rb.hi[c]) & !(rb.lo[c] > ra.hi[c]);
                    ok &= (max(ra.c[0], rb.c[0]) == cx) & (max(ra.c[1], rb.c[1]) == cy) & (max(ra.c[2], rb.c[2]) == cz);
                    ok &= (ra.n0 != rb.n0) & (ra.n0 != rb.n1) & (ra.n1 != rb.n0) & (ra.n1 != rb.n1);
                    const unsigned long long m = __ballot(ok);
                    if (!m) continue;
                    if (ok) q[qn + __popcll(m & ((1ull << lane) - 1ull))] = make_int2(min(ra.id, rb.id), max(ra.id, rb.id));
                    qn += __popcll(m);
                    __builtin_amdgcn_wave_barrier();
                    if (qn >= 64) { // a full wave of pairs: type them, keep the rest
                        const int2 pr = q[lane];
                        const int rest = qn - 64;
                        const int2 mv = q[64 + (lane < rest ? lane : 0)];
                        __builtin_amdgcn_wave_barrier();
                        if (lane < rest) q[lane] = mv;
                        qn = rest;
                        __builtin_amdgcn_wave_barrier();
                        narrow_ee_queued(pr.x, pr.y, SFE, x, xRest, dbc, nE, dHat, wl, cap, out, counter);
                    }
                }
            }
        }
    }
    if (lane < qn) {
        const int2 pr = q[lane];
        narrow_ee_queued(pr.x, pr.y, SFE, x, xRest, dbc, nE, dHat, wl, cap, out, counter);
    }
    wg_list_flush(wl, &sBase, cap, out, counter);
}
// ---- conservative CCD (advancement on the unclassified distance until it meets the gap; contract in DESIGN.md) ---------------
// Converged to ADVANCE_TOL * (initial distance): a bound that stops one step short of the gap jumps by up to a fifth of itself
// when the iteration count changes, and the Newton path that leans on it would not be reproducible.
#define ADVANCE_TOL 1.0e-8
__device__ inline double accd(int kind, const double (*X0)[3], const double (*P0)[3], double eta, double tmax)
{
    double X[4][3], P[4][3], mean[3] = { 0.0, 0.0, 0.0 }, len[4];
    for (int k = 0; k < 4; ++k)
        for (int c = 0; c < 3; ++c) mean[c] += P0[k][c];
    for (int c = 0; c < 3; ++c) mean[c] /= 4.0;
    for (int k = 0; k < 4; ++k) {
        for (int c = 0; c < 3; ++c) {
            X[k][c] = X0[k][c];
            P[k][c] = P0[k][c] - mean[c];
        }
        len[k] = sqrt(dot3(P[k], P[k]));
    }
    const double lp = (kind == K_PT) ? len[0] + fmax(len[1], fmax(len[2], len[3])) : fmax(len[0], len[1]) + fmax(len[2], len[3]);
    if (lp == 0.0) return tmax;
    double d = sqrt(kind == K_PT ? dist2_PT(X[0], X[1], X[2], X[3]) : dist2_EE(X[0], X[1], X[2], X[3]));
    const double gap = eta * d;
    double toc = 0.0;
    const double tol = ADVANCE_TOL * d;
    for (int it = 0; it < 100000; ++it) {
        // the distance cannot shrink faster than lp per unit of t: advancing by (d - gap) / lp never passes d = gap, and the
        // iteration converges onto the first time the distance equals the gap -- what CTCD's thickened query returns
        if (!(d - gap > tol)) break;
        const double tl = (d - gap) / lp;
        toc += tl;
        if (toc > tmax) return tmax;
        for (int k = 0; k < 4; ++k)
            for (int c = 0; c < 3; ++c) X[k][c] += tl * P[k][c];
        d = sqrt(kind == K_PT ? dist2_PT(X[0], X[1], X[2], X[3]) : dist2_EE(X[0], X[1], X[2], X[3]));
    }
    return toc;
}
// time bound of one pair with the retry rule of SelfCollisionHandler.cpp:617-636
__device__ inline double pair_toc(int kind, const int* node, const double* x, const double* p, double slackness, double tmax)
{
    double X[4][3], P[4][3];
    for (int k = 0; k < 4; ++k)
        for (int c = 0; c < 3; ++c) {
            X[k][c] = x[3 * (size_t)node[k] + c];
            P[k][c] = p[3 * (size_t)node[k] + c];
        }
    double t = accd(kind, X, P, 1.0 - slackness, tmax);
    if (t < tmax && t < 1.0e-6) { // asked again almost without safety distance over CTCD's own window [0, 1]: no hit drops the pair
        const double t2 = accd(kind, X, P, 0.01, 1.0);
        t = t2 < 1.0 ? slackness * t2 : tmax;
    }
    return t;
}
// order key of a pair inside the serial enumeration (PT by (svI, sfI), then EE by (eI, eJ)): ties resolve to the first
__device__ __forceinline__ unsigned long long pair_key(int kind, int i, int j)
{
    return ((unsigned long long)(kind == K_PT ? 0 : 1) << 62) | ((unsigned long long)(unsigned)i << 31) | (unsigned long long)(unsigned)j;
}
struct CcdOut {
    unsigned long long* minBits; // bit pattern of the smallest time (positive doubles order like integers)
    unsigned long long* argKey; // smallest order key among the pairs that attain it
    // pairs that returned a time below the step (few: most candidates of a sweep do not meet inside it) as (time bits, order key): the pair
    // that attains the minimum is then found by one small pass over this list instead of a second run of the whole sweep
    unsigned long long* hits = nullptr;
    int* hitCount = nullptr;
    int hitCap = 0;
};
__device__ __forceinline__ void ccd_record_min(double t, double tmax, CcdOut o)
{
    if (t < tmax) atomicMin(o.minBits, (unsigned long long)__double_as_longlong(t));
}
__device__ __forceinline__ void ccd_record_arg(double t, unsigned long long key, CcdOut o)
{
    if ((unsigned long long)__double_as_longlong(t) == *o.minBits) atomicMin(o.argKey, key);
}
// pass 0 of a sweep that keeps a hit list: the minimum, and the pair for the pass over the list
__device__ __forceinline__ void ccd_record_hit(double t, double tmax, unsigned long long key, CcdOut o)
{
    if (!(t < tmax)) return;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(t);
    atomicMin(o.minBits, bits);
    if (o.hits) {
        const int slot = atomicAdd(o.hitCount, 1);
        if (slot < o.hitCap) {
            o.hits[2 * (size_t)slot] = bits;
            o.hits[2 * (size_t)slot + 1] = key;
        }
    }
}
__global__ __launch_bounds__(BLOCK) void k_ccd_hits_arg(CcdOut o)
{
    const int n = min(*o.hitCount, o.hitCap);
    const unsigned long long best = *o.minBits;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK)
        if (o.hits[2 * (size_t)i] == best) atomicMin(o.argKey, o.hits[2 * (size_t)i + 1]);
}

// pass 0: minimum time; pass 1: first pair attaining it
__global__ __launch_bounds__(BLOCK) void k_ccd_list(int nPairs, const int* __restrict__ pairs, const int* __restrict__ SVI, const int* __restrict__ SF,
    const int* __restrict__ SFE, const double* __restrict__ x, const double* __restrict__ p, double slackness, double tmax, int pass, CcdOut o,
    const unsigned long long* __restrict__ tmaxDev)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nPairs) return;
    if (tmaxDev) tmax = __longlong_as_double((long long)tmaxDev[0]); // the bound the inversion filter left on the device (HipContact::stepBounds)
    const int a = pairs[2 * (size_t)i], b = pairs[2 * (size_t)i + 1];
    int node[4], kind, ki, kj;
    if (a < 0) {
        kind = K_PT;
        ki = -a - 1;
        kj = b;
        node[0] = SVI[ki];
        node[1] = SF[3 * (size_t)b];
        node[2] = SF[3 * (size_t)b + 1];
        node[3] = SF[3 * (size_t)b + 2];
    }
    else {
        kind = K_EE;
        ki = a;
        kj = b;
        node[0] = SFE[2 * (size_t)a];
        node[1] = SFE[2 * (size_t)a + 1];
        node[2] = SFE[2 * (size_t)b];
        node[3] = SFE[2 * (size_t)b + 1];
    }
    const double t = pair_toc(kind, node, x, p, slackness, tmax);
    if (pass == 0) ccd_record_min(t, tmax, o);
    else ccd_record_arg(t, pair_key(kind, ki, kj), o);
}

// swept boxes over [x, x + alpha p]
__global__ __launch_bounds__(BLOCK) void k_grid_insert_swept(int nPrim, int nv, const int* __restrict__ prim, const double* __restrict__ x,
    const double* __restrict__ p, double alpha, Grid g, int mode, int* __restrict__ cellCount, const int* __restrict__ cellStart,
    int* __restrict__ cellItems)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nPrim) return;
    double bl[3] = { 1e300, 1e300, 1e300 }, bh[3] = { -1e300, -1e300, -1e300 };
    for (int k = 0; k < nv; ++k) {
        const int v = prim[nv * (size_t)i + k];
        for (int c = 0; c < 3; ++c) {
            const double a = x[3 * (size_t)v + c], b = a + alpha * p[3 * (size_t)v + c];
            bl[c] = fmin(bl[c], fmin(a, b));
            bh[c] = fmax(bh[c], fmax(a, b));
        }
    }
    int a[3], b[3];
    for (int c = 0; c < 3; ++c) {
        a[c] = cell_of(g, bl[c], c);
        b[c] = cell_of(g, bh[c], c);
    }
    for (int z = a[2]; z <= b[2]; ++z)
        for (int y = a[1]; y <= b[1]; ++y)
            for (int xx = a[0]; xx <= b[0]; ++xx) {
                const int cell = xx + g.dim[0] * (y + g.dim[1] * z);
                const int slot = atomicAdd(&cellCount[cell], 1);
                if (mode == 1) cellItems[cellStart[cell] + slot] = i;
            }
}
__device__ __forceinline__ void swept_box(const int* node, int n, const double* x, const double* p, double alpha, double* lo, double* hi)
{
    for (int c = 0; c < 3; ++c) {
        lo[c] = 1e300;
        hi[c] = -1e300;
    }
    for (int k = 0; k < n; ++k)
        for (int c = 0; c < 3; ++c) {
            const double a = x[3 * (size_t)node[k] + c], b = a + alpha * p[3 * (size_t)node[k] + c];
            lo[c] = fmin(lo[c], fmin(a, b));
            hi[c] = fmax(hi[c], fmax(a, b));
        }
}
// full CCD: every (surface vertex, triangle) and (edge, edge) pair with overlapping swept boxes (SelfCollisionHandler.cpp:982-1366)
__global__ __launch_bounds__(BLOCK) void k_ccd_full_pt(int nSVI, const int* __restrict__ SVI, const int* __restrict__ SF, const double* __restrict__ x,
    const double* __restrict__ p, const int* __restrict__ dbc, Grid g, const int* __restrict__ cellStart, const int* __restrict__ cellItems,
    double alpha, double slackness, int pass, CcdOut o, int* __restrict__ nCand)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nSVI) return;
    const int vI = SVI[i];
    double lo[3], hi[3];
    swept_box(&vI, 1, x, p, alpha, lo, hi);
    int ca[3], cb[3];
    for (int c = 0; c < 3; ++c) {
        ca[c] = cell_of(g, lo[c], c);
        cb[c] = cell_of(g, hi[c], c);
    }
    const bool vDbc = (dbc[vI] & 1) != 0;
    for (int z = ca[2]; z <= cb[2]; ++z)
        for (int y = ca[1]; y <= cb[1]; ++y)
            for (int xx = ca[0]; xx <= cb[0]; ++xx) {
                const int cell = xx + g.dim[0] * (y + g.dim[1] * z);
                for (int k = cellStart[cell]; k < cellStart[cell + 1]; ++k) {
                    const int f = cellItems[k];
                    int node[4] = { vI, SF[3 * (size_t)f], SF[3 * (size_t)f + 1], SF[3 * (size_t)f + 2] };
                    if (vI == node[1] || vI == node[2] || vI == node[3]) continue;
                    if (vDbc && (dbc[node[1]] & 1) && (dbc[node[2]] & 1) && (dbc[node[3]] & 1)) continue;
                    if (pair_filtered(dbc[vI], dbc[node[1]])) continue;
                    double tl[3], th[3];
                    swept_box(node + 1, 3, x, p, alpha, tl, th);
                    bool ok = true;
                    int canon[3];
                    for (int c = 0; c < 3; ++c) {
                        if (lo[c] > th[c] || tl[c] > hi[c]) ok = false;
                        canon[c] = cell_of(g, fmax(lo[c], tl[c]), c);
                    }
                    if (!ok || canon[0] != xx || canon[1] != y || canon[2] != z) continue;
                    if (pass == 0) atomicAdd(nCand, 1);
                    const double t = pair_toc(K_PT, node, x, p, slackness, alpha);
                    if (pass == 0) ccd_record_min(t, alpha, o);
                    else ccd_record_arg(t, pair_key(K_PT, i, f), o);
                }
            }
}
__global__ __launch_bounds__(BLOCK) void k_ccd_full_ee(int nE, const int* __restrict__ SFE, const double* __restrict__ x, const double* __restrict__ p,
    const int* __restrict__ dbc, Grid g, const int* __restrict__ cellStart, const int* __restrict__ cellItems, double alpha, double slackness,
    int pass, CcdOut o, int* __restrict__ nCand)
{
    const int eI = blockIdx.x * BLOCK + threadIdx.x;
    if (eI >= nE) return;
    int node[4] = { SFE[2 * (size_t)eI], SFE[2 * (size_t)eI + 1], 0, 0 };
    double lo[3], hi[3];
    swept_box(node, 2, x, p, alpha, lo, hi);
    int ca[3], cb[3];
    for (int c = 0; c < 3; ++c) {
        ca[c] = cell_of(g, lo[c], c);
        cb[c] = cell_of(g, hi[c], c);
    }
    const bool aDbc = (dbc[node[0]] & 1) && (dbc[node[1]] & 1);
    for (int z = ca[2]; z <= cb[2]; ++z)
        for (int y = ca[1]; y <= cb[1]; ++y)
            for (int xx = ca[0]; xx <= cb[0]; ++xx) {
                const int cell = xx + g.dim[0] * (y + g.dim[1] * z);
                for (int k = cellStart[cell]; k < cellStart[cell + 1]; ++k) {
                    const int eJ = cellItems[k];
                    if (eJ <= eI) continue;
                    node[2] = SFE[2 * (size_t)eJ];
                    node[3] = SFE[2 * (size_t)eJ + 1];
                    if (node[0] == node[2] || node[0] == node[3] || node[1] == node[2] || node[1] == node[3]) continue;
                    if (aDbc && (dbc[node[2]] & 1) && (dbc[node[3]] & 1)) continue;
                    if (pair_filtered(dbc[node[0]], dbc[node[2]])) continue;
                    double jl[3], jh[3];
                    swept_box(node + 2, 2, x, p, alpha, jl, jh);
                    bool ok = true;
                    int canon[3];
                    for (int c = 0; c < 3; ++c) {
                        if (lo[c] > jh[c] || jl[c] > hi[c]) ok = false;
                        canon[c] = cell_of(g, fmax(lo[c], jl[c]), c);
                    }
                    if (!ok || canon[0] != xx || canon[1] != y || canon[2] != z) continue;
                    if (pass == 0) atomicAdd(nCand, 1);
                    const double t = pair_toc(K_EE, node, x, p, slackness, alpha);
                    if (pass == 0) ccd_record_min(t, alpha, o);
                    else ccd_record_arg(t, pair_key(K_EE, eI, eJ), o);
                }
            }
}

// ---- full CCD as the reference sweeps it (SelfCollisionHandler.cpp:982-1366 over SpatialHash.hpp:589-832) ------------------------
// The step is capped by the hash first, candidates are the primitives
// whose index boxes in the reference's voxel grid (cell = avgEdgeLen / 3, corner = min over the nodes now and the surface nodes at
// alpha) intersect, a surface vertex is swept against vertices (svJ > svI), edges and triangles, an edge against edges (eJ > eI)
// whose boxes swept over the bound left by the vertex sweeps overlap.  The search grid is the reference's grid coarsened by an
// integer factor m (search cell = reference index / m) so that its size stays bounded; acceptance is decided on the reference
// indices, so m does not change the result.
struct RefGrid {
    double lb[3], oneDiv;
    int m, dim[3];
};
__device__ __forceinline__ int ref_index(const RefGrid& g, double x, int c) { return (int)floor(__dmul_rn(__dsub_rn(x, g.lb[c]), g.oneDiv)); }
__device__ __forceinline__ double swept_pos(double x, double alpha, double p) { return __dadd_rn(x, __dmul_rn(alpha, p)); }

// surface nodes of a mesh collision object (obst flag) are left out and counted out: SpatialHash.hpp:603-618 runs over mesh.SVI, and
// Mesh<3> does not contain them in the reference.  partial[b] = sum, partial[gridDim.x + b] = number of nodes that counted.
__global__ __launch_bounds__(BLOCK) void k_ref_abs_sum(int n, const int* __restrict__ SVI, const int* __restrict__ obst, const double* __restrict__ p,
    double* __restrict__ partial)
{
    __shared__ double sm[BLOCK];
    __shared__ int cnt[BLOCK];
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    double s = 0.0;
    int own = 0;
    if (i < n) {
        const size_t v = (size_t)SVI[i];
        if (!(obst && obst[v])) {
            own = 1;
            s = fabs(p[3 * v]) + fabs(p[3 * v + 1]) + fabs(p[3 * v + 2]);
        }
    }
    cnt[threadIdx.x] = own;
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int off = BLOCK / 2; off > 0; off >>= 1) { // fixed tree: the same bits on every run
        if ((int)threadIdx.x < off) {
            sm[threadIdx.x] += sm[threadIdx.x + off];
            cnt[threadIdx.x] += cnt[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = sm[0];
        partial[gridDim.x + blockIdx.x] = (double)cnt[0];
    }
}
// bounding box of the surface nodes at x + alpha p, one partial per block (6 doubles: lo, hi)
// The prologue of the reference sweep on the device (round 6; three partial arrays used to travel to the host one after the other, each behind a synchronisation):
// k_ref_cap adds the partial sums up in block order -- one thread, the same order as the host loop it replaces, so the cap is bit for bit the same -- and leaves the
// capped step in out[0]; k_ref_bbox_swept_dev reads it from there; k_ref_box_final reduces the two families of partial boxes into out[1 .. 6] (corner, extent) and
// publishes out[0 .. 6] to mapped host memory.
__global__ void k_ref_cap(int nb, const double* __restrict__ partial, double alpha, double voxelSize, double* __restrict__ out)
{
    if (threadIdx.x != 0) return;
    double pSize = 0.0, nOwn = 0.0;
    for (int b = 0; b < nb; ++b) {
        pSize += partial[b];
        nOwn += partial[nb + b];
    }
    pSize /= fmax(nOwn, 1.0) * 3;
    const double spanSize = alpha * pSize / voxelSize; // SpatialHash.hpp:603-618
    if (spanSize > 1) alpha /= spanSize;
    out[0] = alpha;
}
__global__ __launch_bounds__(BLOCK) void k_ref_bbox_swept_dev(int n, const int* __restrict__ SVI, const double* __restrict__ x, const double* __restrict__ p,
    const double* __restrict__ alphaPtr, double* __restrict__ partial)
{
    __shared__ double sm[6][BLOCK / 64];
    const double alpha = alphaPtr[0];
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    if (i < n) {
        const size_t v = (size_t)SVI[i];
        for (int c = 0; c < 3; ++c) lo[c] = hi[c] = swept_pos(x[3 * v + c], alpha, p[3 * v + c]);
    }
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[c] = fmin(lo[c], __shfl_down(lo[c], off, 64));
            hi[c] = fmax(hi[c], __shfl_down(hi[c], off, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            sm[c][threadIdx.x >> 6] = lo[c];
            sm[3 + c][threadIdx.x >> 6] = hi[c];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        double r = sm[threadIdx.x][0];
        for (int k = 1; k < BLOCK / 64; ++k) r = (threadIdx.x < 3) ? fmin(r, sm[threadIdx.x][k]) : fmax(r, sm[threadIdx.x][k]);
        partial[6 * (size_t)blockIdx.x + threadIdx.x] = r;
    }
}
// min / max over nA + nB partial boxes (6 doubles each: lo[3], hi[3]) -> out[1 .. 6]; out[0 .. 6] -> mapped host memory
__global__ __launch_bounds__(BLOCK) void k_ref_box_final(int nA, const double* __restrict__ partA, int nB, const double* __restrict__ partB, double* __restrict__ out,
    double* __restrict__ mapped)
{
    __shared__ double sm[6][BLOCK / 64];
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    for (int b = threadIdx.x; b < nA + nB; b += BLOCK) {
        const double* q = b < nA ? partA + 6 * (size_t)b : partB + 6 * (size_t)(b - nA);
        for (int c = 0; c < 3; ++c) {
            lo[c] = fmin(lo[c], q[c]);
            hi[c] = fmax(hi[c], q[3 + c]);
        }
    }
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[c] = fmin(lo[c], __shfl_down(lo[c], off, 64));
            hi[c] = fmax(hi[c], __shfl_down(hi[c], off, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            sm[c][threadIdx.x >> 6] = lo[c];
            sm[3 + c][threadIdx.x >> 6] = hi[c];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        double r = sm[threadIdx.x][0];
        for (int k = 1; k < BLOCK / 64; ++k) r = (threadIdx.x < 3) ? fmin(r, sm[threadIdx.x][k]) : fmax(r, sm[threadIdx.x][k]);
        out[1 + threadIdx.x] = r;
        mapped[1 + threadIdx.x] = r;
    }
    if (threadIdx.x == 6) mapped[0] = out[0];
}
// index box of every surface vertex: [min(now, then), max(now, then)] per axis (svMinVAI / svMaxVAI, SpatialHash.hpp:640-660)
__global__ __launch_bounds__(BLOCK) void k_ref_vbox(int n, const int* __restrict__ SVI, const double* __restrict__ x, const double* __restrict__ p, double alpha,
    RefGrid g, int* __restrict__ vbox)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const size_t v = (size_t)SVI[i];
    for (int c = 0; c < 3; ++c) {
        const int a = ref_index(g, x[3 * v + c], c), b = ref_index(g, swept_pos(x[3 * v + c], alpha, p[3 * v + c]), c);
        vbox[6 * (size_t)i + c] = min(a, b);
        vbox[6 * (size_t)i + 3 + c] = max(a, b);
    }
}
__device__ __forceinline__ void ref_box(const int* node, int n, const int* __restrict__ v2sv, const int* __restrict__ vbox, int* b)
{
    for (int c = 0; c < 3; ++c) {
        b[c] = 0x7fffffff;
        b[3 + c] = -0x7fffffff;
    }
    for (int k = 0; k < n; ++k) {
        const int* vb = vbox + 6 * (size_t)v2sv[node[k]];
        for (int c = 0; c < 3; ++c) {
            b[c] = min(b[c], vb[c]);
            b[3 + c] = max(b[3 + c], vb[3 + c]);
        }
    }
}
__device__ __forceinline__ bool ref_share(const int* a, const int* b)
{
    return !(a[0] > b[3] || b[0] > a[3] || a[1] > b[4] || b[1] > a[4] || a[2] > b[5] || b[2] > a[5]);
}
// the one search cell in which a pair of intersecting boxes is processed
__device__ __forceinline__ bool ref_canon(const RefGrid& g, const int* a, const int* b, int cx, int cy, int cz)
{
    return max(a[0], b[0]) / g.m == cx && max(a[1], b[1]) / g.m == cy && max(a[2], b[2]) / g.m == cz;
}
// nv nodes per primitive (1: surface vertices given by SVI, 2: edges, 3: triangles); mode 0 counts, mode 1 fills
__global__ __launch_bounds__(BLOCK) void k_ref_insert(int nPrim, int nv, const int* __restrict__ prim, const int* __restrict__ v2sv, const int* __restrict__ vbox,
    RefGrid g, int mode, int* __restrict__ cellCount, const int* __restrict__ cellStart, int* __restrict__ cellItems)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nPrim) return;
    int node[3] = { 0, 0, 0 }, b[6];
    for (int k = 0; k < nv; ++k) node[k] = prim[nv * (size_t)i + k];
    ref_box(node, nv, v2sv, vbox, b);
    for (int z = b[2] / g.m; z <= b[5] / g.m; ++z)
        for (int y = b[1] / g.m; y <= b[4] / g.m; ++y)
            for (int xx = b[0] / g.m; xx <= b[3] / g.m; ++xx) {
                const int cell = xx + g.dim[0] * (y + g.dim[1] * z);
                if (mode == 0) atomicAdd(&cellCount[cell], 1);
                else { // the fill takes its slots by counting the cell back down: the counters are zero again afterwards, no clearing between sweeps
                    const int slot = atomicSub(&cellCount[cell], 1) - 1;
                    int4* r = reinterpret_cast<int4*>(cellItems) + 2 * (size_t)(cellStart[cell] + slot);
                    r[0] = make_int4(i, b[0], b[1], b[2]);
                    r[1] = make_int4(b[3], b[4], b[5], 0);
                }
            }
}
// point-point (n = 2) / point-segment (n = 3) advancement: the scheme of accd() on the point-point / point-segment distance
__device__ inline double dist_ps(const double* p, const double* a, const double* b)
{
    double ab[3], ap[3], abab = 0.0, apab = 0.0;
    for (int c = 0; c < 3; ++c) {
        ab[c] = b[c] - a[c];
        ap[c] = p[c] - a[c];
        abab += ab[c] * ab[c];
        apab += ap[c] * ab[c];
    }
    double s = abab > 0.0 ? apab / abab : 0.0;
    s = s < 0.0 ? 0.0 : (s > 1.0 ? 1.0 : s);
    double d2 = 0.0;
    for (int c = 0; c < 3; ++c) {
        const double r = ap[c] - s * ab[c];
        d2 += r * r;
    }
    return sqrt(d2);
}
__device__ inline double accd_small(int n, const double (*X0)[3], const double (*P0)[3], double eta, double tmax)
{
    double X[3][3], P[3][3], mean[3] = { 0.0, 0.0, 0.0 }, len[3] = { 0.0, 0.0, 0.0 };
    for (int k = 0; k < n; ++k)
        for (int c = 0; c < 3; ++c) mean[c] += P0[k][c];
    for (int c = 0; c < 3; ++c) mean[c] /= (double)n;
    for (int k = 0; k < n; ++k) {
        for (int c = 0; c < 3; ++c) {
            X[k][c] = X0[k][c];
            P[k][c] = P0[k][c] - mean[c];
        }
        len[k] = sqrt(dot3(P[k], P[k]));
    }
    const double lp = n == 2 ? len[0] + len[1] : len[0] + fmax(len[1], len[2]);
    if (lp == 0.0) return tmax;
    const int kb = n == 2 ? 1 : 2;
    double d = dist_ps(X[0], X[1], X[kb]);
    const double gap = eta * d;
    double toc = 0.0;
    const double tol = ADVANCE_TOL * d;
    for (int it = 0; it < 100000; ++it) {
        if (!(d - gap > tol)) break;
        const double tl = (d - gap) / lp;
        toc += tl;
        if (toc > tmax) return tmax;
        for (int k = 0; k < n; ++k)
            for (int c = 0; c < 3; ++c) X[k][c] += tl * P[k][c];
        d = dist_ps(X[0], X[1], X[kb]);
    }
    return toc;
}
// one pair of the reference sweep with its retry rule; kind K_PP / K_PE / K_PT / K_EE
__device__ inline double ref_pair_bound(int kind, const int* node, const double* x, const double* p, double slackness, double tmax)
{
    if (kind == K_PT || kind == K_EE) return pair_toc(kind, node, x, p, slackness, tmax);
    const int n = kind == K_PP ? 2 : 3;
    double X[3][3], P[3][3];
    for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 3; ++c) {
            const int kk = k < n ? k : n - 1;
            X[k][c] = x[3 * (size_t)node[kk] + c];
            P[k][c] = p[3 * (size_t)node[kk] + c];
        }
    double t = accd_small(n, X, P, 1.0 - slackness, tmax);
    if (t < tmax && t < 1.0e-6) {
        const double t2 = accd_small(n, X, P, 0.01, 1.0);
        t = t2 < 1.0 ? slackness * t2 : tmax;
    }
    return t;
}
// order of the serial enumeration: per surface vertex its vertices, edges, triangles; then the edge pairs
__device__ __forceinline__ unsigned long long ref_key(int isEE, int i, int rank, int j)
{
    return ((unsigned long long)isEE << 63) | ((unsigned long long)(unsigned)i << 33) | ((unsigned long long)rank << 31) | (unsigned long long)(unsigned)j;
}
struct RefLists {
    const int *startV, *itemsV, *startE, *itemsE, *startT, *itemsT;
};
// one candidate pair of a sweep: its time bound, the minimum, the hit list (pass 0) or the first pair attaining the minimum (pass 1).  The
// number of queried pairs is kept in a register and added up once per wave at the end of the kernel: bumped per candidate, a global counter
// serialised 3e5 same-address atomics -- 0.8 ms of the sweep's 0.9, whatever the walk did.
__device__ __forceinline__ void ref_pair(int kind, const int* node, unsigned long long key, const double* __restrict__ x, const double* __restrict__ p,
    double slackness, double alpha, int pass, CcdOut o, int& nQueried)
{
    if (pass == 0) ++nQueried;
    const double t = ref_pair_bound(kind, node, x, p, slackness, alpha);
    if (pass == 0) ccd_record_hit(t, alpha, key, o);
    else ccd_record_arg(t, key, o);
}
__device__ __forceinline__ void wave_count_add(int v, int* __restrict__ counter) // every lane of the wave must call it
{
#pragma unroll
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(counter, v);
}
__global__ __launch_bounds__(BLOCK) void k_ref_sweep_vertex(int nSVI, const int* __restrict__ SVI, const int* __restrict__ SF, const int* __restrict__ SFE,
    const double* __restrict__ x, const double* __restrict__ p, const int* __restrict__ pf, const int* __restrict__ v2sv, const int* __restrict__ vbox,
    RefGrid g, RefLists L, double alpha, double slackness, int pass, CcdOut o, int* __restrict__ nCand)
{
    const int gi = blockIdx.x * BLOCK + threadIdx.x;
    const bool valid = gi / SWEEP_COOP < nSVI; // lanes past the end walk an empty range: every lane reaches the wave sum at the end
    const int i = valid ? gi / SWEEP_COOP : 0, sub = gi % SWEEP_COOP; // SWEEP_COOP lanes share a vertex (see COOP)
    const int vI = SVI[i];
    const int* bi = vbox + 6 * (size_t)i;
    const bool vDbc = (pf[vI] & 1) != 0;
    int nQueried = 0;
    for (int z = bi[2] / g.m, zEnd = valid ? bi[5] / g.m : -1; z <= zEnd; ++z)
        for (int y = bi[1] / g.m; y <= bi[4] / g.m; ++y)
            for (int xx = bi[0] / g.m; xx <= bi[3] / g.m; ++xx) {
                const int cell = xx + g.dim[0] * (y + g.dim[1] * z);
                // vertices svJ > svI
                for (int k = L.startV[cell] + sub, kEnd = L.startV[cell + 1]; k < kEnd; k += SWEEP_COOP) {
                    const VoxRec rec = load_vox_rec(L.itemsV, k);
                    const int j = rec.id;
                    if (j <= i) continue;
                    const int* bj = rec.b;
                    if (!ref_share(bi, bj) || !ref_canon(g, bi, bj, xx, y, z)) continue;
                    const int vJ = SVI[j];
                    if (vDbc && (pf[vJ] & 1)) continue;
                    if (pair_filtered(pf[vI], pf[vJ])) continue;
                    const int node[4] = { vI, vJ, vJ, vJ };
                    ref_pair(K_PP, node, ref_key(0, i, 0, j), x, p, slackness, alpha, pass, o, nQueried);
                }
                // edges that do not contain the vertex
                for (int k = L.startE[cell] + sub, kEnd = L.startE[cell + 1]; k < kEnd; k += SWEEP_COOP) {
                    const VoxRec rec = load_vox_rec(L.itemsE, k);
                    const int e = rec.id;
                    const int* be = rec.b; // the edge's voxel box as the insertion computed it
                    if (!ref_share(bi, be) || !ref_canon(g, bi, be, xx, y, z)) continue;
                    const int node[4] = { vI, SFE[2 * (size_t)e], SFE[2 * (size_t)e + 1], SFE[2 * (size_t)e + 1] };
                    if (node[1] == vI || node[2] == vI) continue;
                    if (vDbc && (pf[node[1]] & 1) && (pf[node[2]] & 1)) continue;
                    if (pair_filtered(pf[vI], pf[node[1]])) continue;
                    ref_pair(K_PE, node, ref_key(0, i, 1, e), x, p, slackness, alpha, pass, o, nQueried);
                }
                // triangles that do not contain the vertex
                for (int k = L.startT[cell] + sub, kEnd = L.startT[cell + 1]; k < kEnd; k += SWEEP_COOP) {
                    const VoxRec rec = load_vox_rec(L.itemsT, k);
                    const int f = rec.id;
                    const int* bt = rec.b;
                    if (!ref_share(bi, bt) || !ref_canon(g, bi, bt, xx, y, z)) continue;
                    const int node[4] = { vI, SF[3 * (size_t)f], SF[3 * (size_t)f + 1], SF[3 * (size_t)f + 2] };
                    if (vI == node[1] || vI == node[2] || vI == node[3]) continue;
                    if (vDbc && (pf[node[1]] & 1) && (pf[node[2]] & 1) && (pf[node[3]] & 1)) continue;
                    if (pair_filtered(pf[vI], pf[node[1]])) continue;
                    ref_pair(K_PT, node, ref_key(0, i, 2, f), x, p, slackness, alpha, pass, o, nQueried);
                }
            }
    wave_count_add(nQueried, nCand);
}
// edge pairs eJ > eI that share a cell and whose boxes swept over alphaEE (the bound the vertex sweeps left) overlap
__global__ __launch_bounds__(BLOCK) void k_ref_sweep_edge(int nE, const int* __restrict__ SFE, const double* __restrict__ x, const double* __restrict__ p,
    const int* __restrict__ pf, const int* __restrict__ v2sv, const int* __restrict__ vbox, RefGrid g, const int* __restrict__ startE,
    const int* __restrict__ itemsE, double alpha, const unsigned long long* __restrict__ alphaEEBits, double slackness, int pass, CcdOut o,
    int* __restrict__ nCand)
{
    const int gi = blockIdx.x * BLOCK + threadIdx.x;
    const bool valid = gi / SWEEP_COOP < nE; // see k_ref_sweep_vertex
    const int eI = valid ? gi / SWEEP_COOP : 0, sub = gi % SWEEP_COOP;
    int nQueried = 0;
    double alphaEE = alpha;
    if (*alphaEEBits != ~0ull) alphaEE = fmin(alpha, __longlong_as_double((long long)*alphaEEBits));
    int node[4] = { SFE[2 * (size_t)eI], SFE[2 * (size_t)eI + 1], 0, 0 };
    int bi[6];
    ref_box(node, 2, v2sv, vbox, bi);
    double lo[3], hi[3];
    for (int c = 0; c < 3; ++c) {
        const double a0 = x[3 * (size_t)node[0] + c], a1 = x[3 * (size_t)node[1] + c];
        const double b0 = swept_pos(a0, alphaEE, p[3 * (size_t)node[0] + c]), b1 = swept_pos(a1, alphaEE, p[3 * (size_t)node[1] + c]);
        lo[c] = fmin(fmin(a0, b0), fmin(a1, b1));
        hi[c] = fmax(fmax(a0, b0), fmax(a1, b1));
    }
    const int f0 = pf[node[0]];
    const bool aDbc = (f0 & pf[node[1]] & 1) != 0;
    for (int z = bi[2] / g.m, zEnd = valid ? bi[5] / g.m : -1; z <= zEnd; ++z)
        for (int y = bi[1] / g.m; y <= bi[4] / g.m; ++y)
            for (int xx = bi[0] / g.m; xx <= bi[3] / g.m; ++xx) {
                const int cell = xx + g.dim[0] * (y + g.dim[1] * z);
                for (int k = startE[cell] + sub, kEnd = startE[cell + 1]; k < kEnd; k += SWEEP_COOP) {
                    const VoxRec rec = load_vox_rec(itemsE, k);
                    const int eJ = rec.id;
                    if (eJ <= eI) continue;
                    const int* bj = rec.b;
                    if (!ref_share(bi, bj) || !ref_canon(g, bi, bj, xx, y, z)) continue;
                    node[2] = SFE[2 * (size_t)eJ];
                    node[3] = SFE[2 * (size_t)eJ + 1];
                    // flags, positions and directions requested together, filters without short-circuits (see narrow_ee_queued)
                    const int f2 = pf[node[2]], f3 = pf[node[3]];
                    bool skip = (node[0] == node[2]) | (node[0] == node[3]) | (node[1] == node[2]) | (node[1] == node[3]);
                    for (int c = 0; c < 3; ++c) {
                        const double a0 = x[3 * (size_t)node[2] + c], a1 = x[3 * (size_t)node[3] + c];
                        const double b0 = swept_pos(a0, alphaEE, p[3 * (size_t)node[2] + c]), b1 = swept_pos(a1, alphaEE, p[3 * (size_t)node[3] + c]);
                        const double jl = fmin(fmin(a0, b0), fmin(a1, b1)), jh = fmax(fmax(a0, b0), fmax(a1, b1));
                        skip |= (jl - hi[c] > 0.0) | (lo[c] - jh > 0.0);
                    }
                    skip |= (aDbc & ((f2 & f3 & 1) != 0)) | pair_filtered(f0, f2);
                    if (skip) continue;
                    ref_pair(K_EE, node, ref_key(1, eI, 0, eJ), x, p, slackness, alpha, pass, o, nQueried);
                }
            }
    wave_count_add(nQueried, nCand);
}

// IglUtils::segTriIntersect (IglUtils.hpp:214-265).  exact == 0: the branch the default build compiles (:236-245); exact != 0: the branch of a
// build with USE_PREDICATES (:222-233) -- the segment's ends strictly on opposite sides of the triangle's plane by the exact orientation
// predicate (orient3d_exact.h: floating-point filter, expansion arithmetic behind it).  A template parameter, not an argument: the expansion
// arithmetic would cost the default kernel three quarters of its occupancy (252 VGPRs and 2.7 KB of scratch per lane) without ever running
template <bool exact>
__device__ inline bool seg_tri_intersect(const double* ve0, const double* ve1, const double* vt0, const double* vt1, const double* vt2)
{
    double c0[3], c1[3], c2[3], n[3], r0[3], r1[3], t0[3], t1[3];
    sub3(vt1, vt0, c0);
    sub3(vt2, vt0, c1);
    sub3(ve0, ve1, c2);
    cross3(c0, c1, n);
    sub3(ve0, vt0, r0);
    sub3(ve1, vt0, r1);
    const double det = dot3(n, c2);
    if constexpr (exact) {
        const int o1 = o3::orient3d(vt0, vt1, vt2, ve0), o2 = o3::orient3d(vt0, vt1, vt2, ve1);
        if (o1 == 0 || o2 == 0 || o1 == o2) return false;
    }
    else {
        if (dot3(n, r0) * dot3(n, r1) > 0.0) return false;
        if (det == 0.0) return false;
    }
    // (u, v, t) = coefMtr.fullPivLu().solve(ve0 - vt0), coefMtr = [c0 c1 c2] (IglUtils.hpp:258): Eigen's rank-revealing LU -- pivots below
    // 3 eps |largest pivot| count as zero and the unknowns behind them come out as exactly 0.  Rounds 1-2 used Cramer's rule here: on the flat,
    // obliquely placed sides of a mesh (edge and triangle coplanar up to round-off: det = 1e-20, not 0) it returned ratios of round-off, and
    // one start in twenty was then called intersecting (12_matOnBoard.txt, 5_hitCardHouse.txt were refused); the truncated solve returns t = 0
    // and the in-plane coordinates of the segment's end.  Registers only: rows / columns are swapped by selects, no indexed arrays.
    (void)t0;
    (void)t1;
    double a00 = c0[0], a01 = c1[0], a02 = c2[0], a10 = c0[1], a11 = c1[1], a12 = c2[1], a20 = c0[2], a21 = c1[2], a22 = c2[2];
    double b0 = r0[0], b1 = r0[1], b2 = r0[2];
    int q0 = 0, q1 = 1, q2 = 2; // original column at positions 0, 1, 2
    auto swp = [](double& x, double& y) {
        const double tmp = x;
        x = y;
        y = tmp;
    };
    // ---- step 0: largest entry of the whole matrix, columns first (Eigen's visitor order: the first maximum wins)
    int pr = 0, pc = 0;
    double big = 0.0;
    {
        const double e[9] = { a00, a10, a20, a01, a11, a21, a02, a12, a22 }; // column-major scan
#pragma unroll
        for (int k = 0; k < 9; ++k)
            if (fabs(e[k]) > big) {
                big = fabs(e[k]);
                pr = k % 3;
                pc = k / 3;
            }
    }
    double u = 0.0, v = 0.0, t = 0.0;
    if (big == 0.0) return u >= 0.0 && v >= 0.0 && u + v <= 1.0 && t >= 0.0 && t <= 1.0; // rank 0: the zero vector (as FullPivLU::solve)
    double maxPivot = big;
    if (pr == 1) { swp(a00, a10); swp(a01, a11); swp(a02, a12); swp(b0, b1); }
    if (pr == 2) { swp(a00, a20); swp(a01, a21); swp(a02, a22); swp(b0, b2); }
    if (pc == 1) { swp(a00, a01); swp(a10, a11); swp(a20, a21); const int tq = q0; q0 = q1; q1 = tq; }
    if (pc == 2) { swp(a00, a02); swp(a10, a12); swp(a20, a22); const int tq = q0; q0 = q2; q2 = tq; }
    a10 /= a00;
    a20 /= a00;
    a11 -= a10 * a01;
    a21 -= a20 * a01;
    a12 -= a10 * a02;
    a22 -= a20 * a02;
    // ---- step 1: largest entry of the trailing 2 x 2, columns first
    int nonzero = 3;
    pr = 1;
    pc = 1;
    big = 0.0;
    if (fabs(a11) > big) { big = fabs(a11); pr = 1; pc = 1; }
    if (fabs(a21) > big) { big = fabs(a21); pr = 2; pc = 1; }
    if (fabs(a12) > big) { big = fabs(a12); pr = 1; pc = 2; }
    if (fabs(a22) > big) { big = fabs(a22); pr = 2; pc = 2; }
    if (big == 0.0) nonzero = 1;
    else {
        if (big > maxPivot) maxPivot = big;
        if (pr == 2) { swp(a10, a20); swp(a11, a21); swp(a12, a22); swp(b1, b2); }
        if (pc == 2) { swp(a01, a02); swp(a11, a12); swp(a21, a22); const int tq = q1; q1 = q2; q2 = tq; }
        a21 /= a11;
        a22 -= a21 * a12;
        // ---- step 2
        if (a22 == 0.0) nonzero = 2;
        else if (fabs(a22) > maxPivot) maxPivot = fabs(a22);
    }
    const double thr = maxPivot * (2.220446049250313e-16 * 3.0);
    int rank = 0;
    rank += (nonzero > 0 && fabs(a00) > thr) ? 1 : 0;
    rank += (nonzero > 1 && fabs(a11) > thr) ? 1 : 0;
    rank += (nonzero > 2 && fabs(a22) > thr) ? 1 : 0;
    // unit-lower solve on all three rows, upper solve on the leading rank x rank corner, zeros behind it
    double y0 = b0, y1 = b1 - a10 * y0, y2 = b2 - a20 * y0 - a21 * y1;
    double z0 = 0.0, z1 = 0.0, z2 = 0.0;
    if (rank == 3) {
        z2 = y2 / a22;
        z1 = (y1 - a12 * z2) / a11;
        z0 = ((y0 - a01 * z1) - a02 * z2) / a00;
    }
    else if (rank == 2) {
        z1 = y1 / a11;
        z0 = (y0 - a01 * z1) / a00;
    }
    else if (rank == 1) z0 = y0 / a00;
    (void)y2;
    // x[q_i] = z_i
    u = (q0 == 0) ? z0 : ((q1 == 0) ? z1 : z2);
    v = (q0 == 1) ? z0 : ((q1 == 1) ? z1 : z2);
    t = (q0 == 2) ? z0 : ((q1 == 2) ? z1 : z2);
    return u >= 0.0 && v >= 0.0 && u + v <= 1.0 && t >= 0.0 && t <= 1.0;
}
// checkEdgeTriIntersectionIfAny (SelfCollisionHandler.cpp:3255-3300): one lane per surface triangle, edges from the grid
template <bool exact>
__global__ __launch_bounds__(BLOCK) void k_intersect(int nSF, const int* __restrict__ SF, const int* __restrict__ SFE, const double* __restrict__ x,
    const int* __restrict__ dbc, Grid g, const int* __restrict__ cellStart, const int* __restrict__ cellItems, int* __restrict__ flag, int capItems)
{
    const int gi = blockIdx.x * BLOCK + threadIdx.x;
    const int f = gi / COOP, sub = gi % COOP; // COOP lanes share a triangle and stride over the edges of each cell (see COOP)
    if (f >= nSF) return;
    const int t0 = SF[3 * (size_t)f], t1 = SF[3 * (size_t)f + 1], t2 = SF[3 * (size_t)f + 2];
    double a[3], b[3], c[3], lo[3], hi[3];
    int ca[3], cb[3];
    for (int k = 0; k < 3; ++k) {
        a[k] = x[3 * (size_t)t0 + k];
        b[k] = x[3 * (size_t)t1 + k];
        c[k] = x[3 * (size_t)t2 + k];
        lo[k] = fmin(a[k], fmin(b[k], c[k]));
        hi[k] = fmax(a[k], fmax(b[k], c[k]));
        ca[k] = cell_of(g, lo[k], k);
        cb[k] = cell_of(g, hi[k], k);
    }
    const int ft0 = dbc[t0];
    const bool tDbc = (ft0 & dbc[t1] & dbc[t2] & 1) != 0;
    const float loF[3] = { f_down(lo[0]), f_down(lo[1]), f_down(lo[2]) }, hiF[3] = { f_up(hi[0]), f_up(hi[1]), f_up(hi[2]) };
    for (int z = ca[2]; z <= cb[2]; ++z)
        for (int y = ca[1]; y <= cb[1]; ++y)
            for (int xx = ca[0]; xx <= cb[0]; ++xx) {
                const int cell = xx + g.dim[0] * (y + g.dim[1] * z);
                for (int k = cellStart[cell] + sub, kEnd = min(cellStart[cell + 1], capItems); k < kEnd; k += COOP) { // (capItems: a truncated list -- the host repeats)
                    const BoxRec rec = load_box_rec(cellItems, k);
                    if (rec.lo[0] > hiF[0] || rec.hi[0] < loF[0] || rec.lo[1] > hiF[1] || rec.hi[1] < loF[1] || rec.lo[2] > hiF[2] || rec.hi[2] < loF[2])
                        continue; // outward-rounded boxes apart: the exact test below would say the same
                    const int e = rec.id;
                    const int e0 = SFE[2 * (size_t)e], e1 = SFE[2 * (size_t)e + 1];
                    // most records that get here are the ~15 edges around the triangle's own nodes: they go before anything else is fetched; for the rest, flags and
                    // positions are requested together and the filters evaluated without short-circuits (see narrow_ee_queued)
                    if ((e0 == t0) | (e0 == t1) | (e0 == t2) | (e1 == t0) | (e1 == t1) | (e1 == t2)) continue;
                    const int fe0 = dbc[e0], fe1 = dbc[e1];
                    double p0[3], p1[3];
                    bool skip = false;
                    for (int q = 0; q < 3; ++q) {
                        p0[q] = x[3 * (size_t)e0 + q];
                        p1[q] = x[3 * (size_t)e1 + q];
                        skip |= (fmin(p0[q], p1[q]) > hi[q]) | (fmax(p0[q], p1[q]) < lo[q]);
                    }
                    skip |= (tDbc & ((fe0 & fe1 & 1) != 0)) | pair_filtered(fe0, ft0);
                    if (skip) continue;
                    if (seg_tri_intersect<exact>(p0, p1, a, b, c)) {
                        atomicOr(flag, 1);
                        return;
                    }
                }
            }
}
// codimensional points against every tetrahedron (SelfCollisionHandler.cpp:3301-3338): inside the element's box, then behind its four faces
// (IglUtils::pointInsideTetrahedron / pointBehindTri, IglUtils.hpp:266-311, the build without exact predicates).  One lane per (point, element).
__device__ __forceinline__ bool point_behind_tri(const double* t0, const double* t1, const double* t2, const double* v)
{
    double e1[3], e2[3], r[3], n[3];
    sub3(t1, t0, e1);
    sub3(t2, t0, e2);
    sub3(v, t0, r);
    cross3(e1, e2, n);
    return dot3(n, r) <= 0.0;
}
template <bool exact>
__global__ __launch_bounds__(BLOCK) void k_points_in_tets(int nPts, const int* __restrict__ pts, int nT, const int4* __restrict__ tet,
    const double* __restrict__ x, int* __restrict__ flag)
{
    const long long gi = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (gi >= (long long)nPts * nT) return;
    const int v = pts[gi / nT];
    const int4 t = tet[gi % nT];
    const int id[4] = { t.x, t.y, t.z, t.w };
    double p[3], q[4][3];
    for (int c = 0; c < 3; ++c) {
        p[c] = x[3 * (size_t)v + c];
        for (int k = 0; k < 4; ++k) q[k][c] = x[3 * (size_t)id[k] + c];
        if (!(fmin(fmin(q[0][c], q[1][c]), fmin(q[2][c], q[3][c])) <= p[c] && fmax(fmax(q[0][c], q[1][c]), fmax(q[2][c], q[3][c])) >= p[c])) return;
    }
    if constexpr (exact) { // IglUtils.hpp:280-294: orient3d(...) != NEGATIVE four times
        if (o3::orient3d(q[0], q[2], q[1], p) >= 0 && o3::orient3d(q[0], q[3], q[2], p) >= 0 && o3::orient3d(q[0], q[1], q[3], p) >= 0
            && o3::orient3d(q[1], q[2], q[3], p) >= 0)
            atomicOr(flag, 1);
        return;
    }
    if (point_behind_tri(q[0], q[2], q[1], p) && point_behind_tri(q[0], q[3], q[2], p) && point_behind_tri(q[0], q[1], q[3], p)
        && point_behind_tri(q[1], q[2], q[3], p))
        atomicOr(flag, 1);
}
// squared distances of a list of MMCVID stencils (closeMConstraint bookkeeping, Optimizer.cpp:2365-2440)
__global__ __launch_bounds__(BLOCK) void k_eval_stencils(int n, const int* __restrict__ ids, const double* __restrict__ x, double* __restrict__ out)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const Stencil s = decode(ids + 4 * (size_t)i);
    double X[4][3];
    gatherX(x, s.node, s.n, X);
    out[i] = dist_only(s.kind, X);
}

// ---- constraint-set assembly on the device (SelfCollisionHandler.cpp:2411-2476) ----------------------------------------------
// The narrow phase appends candidate records through an atomic counter, in no particular order.  The reference emits them in
// primitive order (vertex by vertex, edge by edge) and merges the point-point / point-edge duplicates in a std::map keyed by
// the 4-tuple.  Same result here without the host, as two COUNTING sorts (round 6; rounds 2-5 ran radix sorts on 64-bit keys, which rocPRIM turns into a
// merge sort of ~15 launches at these sizes -- 67 launches of ~5 us per Newton iteration):
//   records by primitive pair: the narrow phase counts the records of every first primitive (WgList::bucket), one scan gives the buckets, k_bucket_fill
//   drops every record into its bucket, k_bucket_rank_classify ranks each record inside its bucket by the second primitive (a handful of entries) and classifies;
//   duplicates by tuple: the bucket is the tuple's vertex (k_bucket_rank_classify counts, k_scatter_sets fills), k_dup_rank orders each bucket by the other
//   two ids, k_dup_runs counts its runs, one scan places the runs, k_dup_emit writes one tuple per run with its multiplicity.
// Every counter array is counted up by one kernel and back down to zero by the one that fills the buckets: no clearing between builds.

// one thread per record: its slot in the bucket of its first primitive (point-triangle buckets [0, nSVI), edge-edge buckets behind them)
__global__ void k_bucket_fill(int nPT, int nEE, int nSVI, const int* __restrict__ recPT, const int* __restrict__ recEE, const int* __restrict__ start,
    int* __restrict__ count, int2* __restrict__ seg)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nPT + nEE) return;
    const bool isEE = j >= nPT;
    const int i = isEE ? j - nPT : j;
    const int* r = (isEE ? recEE : recPT) + 6 * (size_t)i;
    const int b = isEE ? nSVI + r[4] : r[4];
    const int slot = start[b] + atomicSub(count + b, 1) - 1;
    seg[slot] = make_int2(r[5], i); // (second primitive, index in its own list)
}
// category of a record: 0 direct active, 1 duplicate candidate (PP / PE), 2 mollified parallel edge pair.  Packed counters for
// one 64-bit prefix sum: bits 0..31 direct, 32..63 duplicate; the parallel pairs in front of record j are the rest, j - direct - duplicate
// (round 6: three 21-bit fields capped a set at 2 M candidate pairs -- the 1.1 M-tet stack has 1.2 M).
__device__ __forceinline__ int rec_category(const int* r, bool isEE)
{
    if (!isEE) return r[3] < 0 ? 1 : 0;
    return r[3] >= 0 ? 0 : (r[3] == -1 ? 1 : 2);
}
// what the host needs between the stages of a build, written to mapped host memory by one thread (three blits of 12, 4 and 48 bytes before)
struct BuildReadback {
    int cnt[4]; // point-triangle records, edge-edge records, stale-grid flag, total number of cell entries
    double box[6];
    unsigned long long totals; // category totals of the candidate list (k_bucket_rank_classify's flags, scanned)
    int nUnique, pad;
};
// one thread per bucket ENTRY: the rank of its second primitive inside the bucket (the pair is unique, the bucket a handful of entries) gives the record's
// place in the serial enumeration -- the bucket's range of the list IS its range there --, and the thread writes what belongs to that place: the permutation
// (point-triangle records first, then the edge-edge ones, each as an index into its own list), the candidate list, the category flag for the prefix sum; it
// also counts the duplicate candidates per vertex
__global__ void k_bucket_rank_classify(int nPT, int nEE, int nSVI, int nV, const int* __restrict__ start, const int2* __restrict__ seg, const int* __restrict__ recPT,
    const int* __restrict__ recEE, int* __restrict__ perm, int* __restrict__ csPTEE, unsigned long long* __restrict__ flags, int* __restrict__ dupCount)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = nPT + nEE;
    if (u == n) flags[n] = 0ull; // the scan runs over n + 1 entries: its last output is the totals
    if (u >= n) return;
    const bool isEE = u >= nPT;
    const int2 e = seg[u];
    const int* r = (isEE ? recEE : recPT) + 6 * (size_t)e.y;
    const int b = isEE ? nSVI + r[4] : r[4];
    const int s0 = start[b], s1 = start[b + 1];
    int rank = 0;
    for (int w = s0; w < s1; ++w) rank += seg[w].x < e.x ? 1 : 0;
    const int s = s0 + rank;
    perm[s] = e.y;
    csPTEE[2 * (size_t)s] = isEE ? r[4] : -r[4] - 1;
    csPTEE[2 * (size_t)s + 1] = r[5];
    const int cat = rec_category(r, isEE);
    flags[s] = cat == 0 ? 1ull : (cat == 1 ? 1ull << 32 : 0ull);
    if (cat == 1) atomicAdd(dupCount + (r[0] + nV), 1); // id0 = -v - 1 in [-nV, -1]: buckets in ascending id0, the map's (signed) order
}
__global__ void k_scatter_sets(int nPT, int nEE, int nSFE, const int* __restrict__ recPT, const int* __restrict__ permPT,
    const int* __restrict__ recEE, const int* __restrict__ permEE, const unsigned long long* __restrict__ pos, int* __restrict__ active,
    int* __restrict__ dupTuple, const int* __restrict__ dupStart, int* __restrict__ dupCount, int* __restrict__ para, int* __restrict__ paraEIEJ, int nV)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nPT + nEE) return;
    const bool isEE = j >= nPT;
    const int* r = isEE ? recEE + 6 * (size_t)permEE[j - nPT] : recPT + 6 * (size_t)permPT[j];
    const unsigned long long p = pos[j];
    const int cat = rec_category(r, isEE);
    const int nd = (int)(p & 0xffffffffull), nu = (int)(p >> 32);
    if (cat == 0) {
        for (int k = 0; k < 4; ++k) active[4 * (size_t)nd + k] = r[k];
    }
    else if (cat == 1) {
        const int v = r[0] + nV;
        const int q = dupStart[v] + atomicSub(dupCount + v, 1) - 1; // any slot of the vertex's bucket: k_dup_rank orders it
        for (int k = 0; k < 4; ++k) dupTuple[4 * (size_t)q + k] = r[k];
    }
    else {
        const int q = j - nd - nu;
        int* o = para + 4 * (size_t)q;
        o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
        if (r[3] >= -nSFE - 1) {
            o[3] = -1;
            paraEIEJ[2 * (size_t)q] = r[4];
            paraEIEJ[2 * (size_t)q + 1] = -r[3] - 2;
        }
        else {
            o[3] = -r[3] - nSFE - 2;
            paraEIEJ[2 * (size_t)q] = -1;
            paraEIEJ[2 * (size_t)q + 1] = -1;
        }
    }
}
// one thread per duplicate candidate: its rank inside its vertex's bucket by (id1, id2) (id2 = -1 of a point-point tuple sorts first, as in the map's signed
// order; id3 = -1 for all of them; equal tuples in the order they happen to lie in -- they are merged anyway)
__global__ void k_dup_rank(const unsigned long long* __restrict__ totals, int nV, const int* __restrict__ start, const int4* __restrict__ tuple,
    int4* __restrict__ sorted)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= (int)(totals[0] >> 32)) return; // the number of duplicate candidates never leaves the device before the build is complete
    const int4 t = tuple[u];
    const int v = t.x + nV;
    const int s0 = start[v], s1 = start[v + 1];
    int rank = 0;
    for (int w = s0; w < s1; ++w) {
        const int4 o = tuple[w];
        rank += (o.y < t.y || (o.y == t.y && (o.z < t.z || (o.z == t.z && w < u)))) ? 1 : 0;
    }
    sorted[s0 + rank] = t;
}
// one thread per vertex: the number of distinct tuples among its (ordered) duplicate candidates
__global__ void k_dup_runs(int nV, const int* __restrict__ start, const int4* __restrict__ tuple, int* __restrict__ runs)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v == 0) runs[nV] = 0; // (the scan's last output is the number of runs)
    if (v >= nV) return;
    const int s0 = start[v], s1 = start[v + 1];
    int nr = s1 > s0 ? 1 : 0;
    for (int a = s0 + 1; a < s1; ++a) nr += (tuple[a].y != tuple[a - 1].y || tuple[a].z != tuple[a - 1].z) ? 1 : 0;
    runs[v] = nr;
}
// one thread per vertex: (tuple, -multiplicity) of each of its runs behind the direct entries
__global__ void k_dup_emit(int nV, const unsigned long long* __restrict__ totals, const int* __restrict__ start, const int4* __restrict__ tuple,
    const int* __restrict__ runPos, int4* __restrict__ active, BuildReadback* __restrict__ out)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v == 0) { // what the host sizes the sets with (mapped memory): category totals and the number of merged tuples
        out->totals = totals[0];
        out->nUnique = runPos[nV];
    }
    if (v >= nV) return;
    const int nDirect = (int)(totals[0] & 0xffffffffull);
    const int s0 = start[v], s1 = start[v + 1];
    if (s0 == s1) return;
    int4* o = active + nDirect + runPos[v];
    int a = s0;
    while (a < s1) {
        int len = 1;
        while (a + len < s1 && tuple[a + len].y == tuple[a].y && tuple[a + len].z == tuple[a].z) ++len;
        *o++ = make_int4(tuple[a].x, tuple[a].y, tuple[a].z, -len);
        a += len;
    }
}
__global__ void k_publish_narrow(const int* __restrict__ counters, const int* __restrict__ gridTotal, const double* __restrict__ box, BuildReadback* __restrict__ out)
{
    out->cnt[0] = counters[0];
    out->cnt[1] = counters[1];
    out->cnt[2] = counters[2];
    out->cnt[3] = gridTotal[0];
    for (int c = 0; c < 6; ++c) out->box[c] = box[c];
}
__global__ void k_publish_u64(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst) { dst[0] = src[0]; }
__global__ void k_publish_u64x4(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst) { dst[threadIdx.x] = src[threadIdx.x]; }
// ccdOut of HipContact::stepBounds: [0] minimum time, [1] its pair (both "none"), [2] largest surface speed, [3] the step the CCD is bounded by
__global__ void k_step_bounds_init(double stepSize, const double* __restrict__ filterRoot, unsigned long long* __restrict__ out)
{
    out[0] = ~0ull;
    out[1] = ~0ull;
    out[2] = 0ull;
    if (filterRoot) {
        const double t = filterRoot[0];
        if (t > 0.0 && t < stepSize) stepSize = t; // Energy.cpp:565-581
    }
    out[3] = (unsigned long long)__double_as_longlong(stepSize);
}
__global__ void k_publish_int(const int* __restrict__ src, int* __restrict__ dst) { dst[0] = src[0]; }
// stencils of the active set closer than dTol (closeMConstraint bookkeeping, Optimizer.cpp:2365-2440): index + distance
__global__ void k_close_stencils(int n, const int* __restrict__ ids, const double* __restrict__ x, double dTol, int cap, int* __restrict__ outIdx,
    double* __restrict__ outVal, int* __restrict__ counter)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const Stencil s = decode(ids + 4 * (size_t)i);
    double X[4][3];
    gatherX(x, s.node, s.n, X);
    const double d = dist_only(s.kind, X);
    if (d < dTol) {
        const int slot = atomicAdd(counter, 1);
        if (slot < cap) {
            outIdx[slot] = i;
            outVal[slot] = d;
        }
    }
}

inline int nblk(long long n, int b = BLOCK) { return (int)((n + b - 1) / b); }

} // namespace

void HipContact::setSurface(const HipMesh& mesh, int nSF_, const int* SFc, int nCE, const int* CE)
{
    nSF = nSF_;
    SF.assign(SFc, SFc + 3 * (size_t)nSF);
    std::set<std::pair<int, int>> es;
    std::vector<char> onSurf(mesh.nV, 0);
    std::vector<int> sfRow(3 * (size_t)nSF);
    for (int f = 0; f < nSF; ++f) {
        const int t[3] = { SFc[f], SFc[f + (size_t)nSF], SFc[f + 2 * (size_t)nSF] };
        for (int k = 0; k < 3; ++k) {
            if (t[k] < 0 || t[k] >= mesh.nV) throw ArgError("set_surface: vertex index out of range");
            onSurf[t[k]] = 1;
            sfRow[3 * (size_t)f + k] = t[k];
        }
        // Mesh.cpp:495-511: an edge is kept unless its reverse is already there
        if (!es.count({ t[1], t[0] })) es.insert({ t[0], t[1] });
        if (!es.count({ t[2], t[1] })) es.insert({ t[1], t[2] });
        if (!es.count({ t[0], t[2] })) es.insert({ t[2], t[0] });
    }
    SFEdges.assign(es.begin(), es.end());
    for (int e = 0; e < nCE; ++e) { // Mesh.cpp:513-515, 912-915: behind the triangles' edges, in file order; the ends are surface vertices
        const int a = CE[2 * (size_t)e], b = CE[2 * (size_t)e + 1];
        if (a < 0 || a >= mesh.nV || b < 0 || b >= mesh.nV) throw ArgError("set_surface: vertex index of a codimensional segment out of range");
        SFEdges.emplace_back(a, b);
        onSurf[a] = onSurf[b] = 1;
    }
    codimPoints.clear();
    for (int v = 0; v < mesh.nV; ++v) // Mesh.cpp:916-920: a node without any neighbour is on the surface (`.pt` shapes)
        if (mesh.nbPtr[v + 1] == mesh.nbPtr[v]) {
            onSurf[v] = 1;
            codimPoints.push_back(v);
        }
    d_codimPoints.upload(codimPoints, stream);
    nSFE = (int)SFEdges.size();
    SVI.clear();
    for (int v = 0; v < mesh.nV; ++v)
        if (onSurf[v]) SVI.push_back(v); // Mesh.cpp:921-927
    nSVI = (int)SVI.size();
    std::vector<int> sfe(2 * (size_t)nSFE);
    for (int e = 0; e < nSFE; ++e) {
        sfe[2 * (size_t)e] = SFEdges[e].first;
        sfe[2 * (size_t)e + 1] = SFEdges[e].second;
    }
    std::vector<int> v2sv((size_t)mesh.nV, -1);
    for (int i = 0; i < nSVI; ++i) v2sv[(size_t)SVI[i]] = i;
    d_v2sv.upload(v2sv, stream);
    d_SF.upload(sfRow, stream);
    d_SVI.upload(SVI, stream);
    d_SFE.upload(sfe, stream);
    std::vector<double> xr(3 * (size_t)mesh.nV);
    for (int v = 0; v < mesh.nV; ++v)
        for (int c = 0; c < 3; ++c) xr[3 * (size_t)v + c] = mesh.V_rest[v + (size_t)mesh.nV * c];
    d_xRest.upload(xr, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
    surfaceSet = true;
    haveBox_ = false;
    active.clear();
    nActive_ = nPara_ = nCand_ = 0;
    hostStale_ = false;
    para.clear();
    paraEIEJ.clear();
    csPTEE.clear();
    uploadSets();
}

// obstacle nodes (MeshCO as a surface-only component) and the "only pairs with an obstacle" filter
void HipContact::setObstacle(int nV, int n, const int* ids, bool only)
{
    std::vector<int> f((size_t)std::max(nV, 1), 0);
    for (int i = 0; i < n; ++i) {
        if (ids[i] < 0 || ids[i] >= nV) throw ArgError("set_obstacle_nodes: node id out of range");
        f[ids[i]] = 1;
    }
    d_obst.upload(f, stream);
    hasObstacle = n > 0;
    obstacleOnly = only;
    HIP_CHECK(hipStreamSynchronize(stream));
}
const int* HipContact::pairFlags(int nV, const int* dbc_dev)
{
    d_pairFlags.ensure((size_t)std::max(nV, 1));
    hipLaunchKernelGGL(k_pair_flags, dim3(nblk(nV)), dim3(BLOCK), 0, stream, nV, dbc_dev, hasObstacle ? d_obst.p : (const int*)nullptr, obstacleOnly ? 1 : 0,
        d_pairFlags.p);
    return d_pairFlags.p;
}

void HipContact::setSets(int nA, const int* a4, int nP, const int* p4, const int* pe2)
{
    active.resize(nA);
    for (int i = 0; i < nA; ++i) active[i] = { a4[4 * i], a4[4 * i + 1], a4[4 * i + 2], a4[4 * i + 3] };
    para.resize(nP);
    paraEIEJ.resize(nP);
    for (int i = 0; i < nP; ++i) {
        para[i] = { p4[4 * i], p4[4 * i + 1], p4[4 * i + 2], p4[4 * i + 3] };
        paraEIEJ[i] = { pe2[2 * i], pe2[2 * i + 1] };
    }
    uploadSets();
}

void HipContact::uploadSets()
{
    std::vector<int> a(4 * std::max<size_t>(1, active.size()), 0), p(4 * std::max<size_t>(1, para.size()), 0),
        q(2 * std::max<size_t>(1, para.size()), 0);
    for (size_t i = 0; i < active.size(); ++i)
        for (int k = 0; k < 4; ++k) a[4 * i + k] = active[i][k];
    for (size_t i = 0; i < para.size(); ++i) {
        for (int k = 0; k < 4; ++k) p[4 * i + k] = para[i][k];
        q[2 * i] = paraEIEJ[i][0];
        q[2 * i + 1] = paraEIEJ[i][1];
    }
    d_active.uploadGrow(a, stream);
    d_para.uploadGrow(p, stream);
    d_paraEIEJ.uploadGrow(q, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
    nActive_ = (int)active.size();
    nPara_ = (int)para.size();
    ++setsVersion;
    hostStale_ = false; // the host vectors are the source here
}

void HipContact::readbackInit()
{
    if (!readback_.p) readback_.alloc(16); // >= sizeof(BuildReadback) / 8
    static_assert(sizeof(BuildReadback) <= 16 * sizeof(unsigned long long), "read-back block too small");
}

#ifndef GRID_H_SCALE
#define GRID_H_SCALE 1.0 // cell size of the narrow phase's grid in mean edge lengths
#endif
int HipContact::buildConstraintSet(const HipMesh& mesh, const double* x_dev, const int* dbc_dev, double dHat)
{
    if (!surfaceSet) throw StateError("contact_build before set_surface");
    ++setsVersion;
    const int nV = mesh.nV;
    const int* pf = pairFlags(nV, dbc_dev);
    const double infl = std::sqrt(dHat);
    // Bounding box -> grid.  cell_of() clamps, so a grid laid over an OLDER box is still a correct grid (whatever lies outside lands in the border cells, on both
    // sides of every pair): the box of this build is measured on the device and read back with the narrow phase's counts, the grid uses the box the PREVIOUS
    // build measured (padded by two cells).  Only the first build of a surface waits for its own box.  (Round 6: one synchronisation per build less; the second --
    // the total number of cell entries -- went the same way: the fill pass writes into the capacity the last build needed and the total comes back with the counts.)
    const int nb = nblk(nV);
    bboxPartial_.ensure(6 * (size_t)nb + 6);
    double* box_dev = bboxPartial_.p + 6 * (size_t)nb;
    counters_.alloc(16);
    const int* stale = counters_.p + 2;
    double boxNow[6];
    if (!haveBox_) {
        counters_.zero(stream);
        hipLaunchKernelGGL(k_bbox_partial, dim3(nb), dim3(BLOCK), 0, stream, nV, x_dev, bboxPartial_.p);
        hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(BLOCK), 0, stream, nb, bboxPartial_.p, box_dev, Grid{}, 0, counters_.p + 2);
        HIP_CHECK(hipMemcpyAsync(boxNow, box_dev, sizeof(boxNow), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        for (int c = 0; c < 6; ++c) box_[c] = boxNow[c];
        haveBox_ = true;
    }
    int capPT = std::max<int>(1 << 14, (int)outPT_.n / 6), capEE = std::max<int>(1 << 14, (int)outEE_.n / 6);
    const int nPrim = nSF + nSFE;
    if (gridItems_.n < (size_t)REC * 16 * (size_t)nPrim) gridItems_.ensure((size_t)REC * 16 * (size_t)nPrim); // (a first guess: 16 cells per primitive)
    // counters of the counting sorts (see "constraint-set assembly on the device"): cleared when allocated and after a build that did not run to its end;
    // a build that does leaves them zero
    const int nB = nSVI + nSFE;
    if (bucketCount_.n < (size_t)nB + 1 || dupCount_.n < (size_t)nV + 1 || countersDirty_) {
        bucketCount_.ensure((size_t)nB + 1);
        dupCount_.ensure((size_t)nV + 1);
        bucketCount_.zero(stream);
        dupCount_.zero(stream);
    }
    countersDirty_ = true;
    bucketStart_.ensure((size_t)nB + 1);
    dupStart_.ensure((size_t)nV + 1);
    runs_.ensure((size_t)nV + 1);
    runPos_.ensure((size_t)nV + 1);
    readbackInit();
    BuildReadback* rb = reinterpret_cast<BuildReadback*>(readback_.p);
    BuildReadback* rbDev = reinterpret_cast<BuildReadback*>(readback_.dev);
    auto scan = [&](auto* in, auto* out, int count) { // exclusive prefix sum, temporary storage grown on demand
        size_t bytes = 0;
        hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, count, stream);
        if (scanTmp_.n < bytes) scanTmp_.alloc(bytes + bytes / 4);
        hipcub::DeviceScan::ExclusiveSum((void*)scanTmp_.p, bytes, in, out, count, stream);
    };
    int nPT = 0, nEE = 0;
    Grid g;
    for (int attempt = 0;; ++attempt) {
        if (attempt > 8) throw StateError("constraint-set build: the grid does not settle");
        g.h = std::max(GRID_H_SCALE * mesh.avgEdgeLen, 2.0 * infl);
        long long nCells;
        for (;;) {
            nCells = 1;
            for (int c = 0; c < 3; ++c) {
                g.lo[c] = box_[c] - 2.0 * g.h;
                g.dim[c] = std::max(1, (int)std::floor((box_[3 + c] + 2.0 * g.h - g.lo[c]) / g.h) + 1);
                nCells *= g.dim[c];
            }
            if (nCells <= (1LL << 26)) break;
            g.h *= 1.5;
        }
        if (2 * nCells + 1 > (long long)INT_MAX) throw StateError("constraint-set build: grid too fine for 32-bit cell indices");
        // both grids in one pass: counters [0, nCells) of the triangles, [nCells, 2 nCells) of the edges, one scan (k_grid_insert_both)
        const size_t nC2 = 2 * (size_t)nCells + 1;
        if (gridCount_.n < nC2) { // a fresh array is cleared once; every pass leaves it zero (the fill counts back down)
            gridCount_.ensure(nC2);
            gridCount_.zeroN(gridCount_.n, stream);
        }
        gridStart_.ensure(nC2);
        const int* cellStartT = gridStart_.p;
        const int* cellStartE = gridStart_.p + nCells;
        outPT_.alloc(6 * (size_t)capPT);
        outEE_.alloc(6 * (size_t)capEE);
        const int capItems = (int)std::min<size_t>(gridItems_.n / REC, (size_t)INT_MAX);
        counters_.zero(stream);
        hipLaunchKernelGGL(k_bbox_partial, dim3(nb), dim3(BLOCK), 0, stream, nV, x_dev, bboxPartial_.p);
        hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(BLOCK), 0, stream, nb, bboxPartial_.p, box_dev, g, 1, counters_.p + 2);
        hipLaunchKernelGGL(k_grid_insert_both, dim3(nblk(nPrim)), dim3(BLOCK), 0, stream, nSF, d_SF.p, nSFE, d_SFE.p, x_dev, g, (int)nCells, infl, 0, capItems, stale,
            gridCount_.p, (const int*)nullptr, (int*)nullptr);
        scan(gridCount_.p, gridStart_.p, (int)nC2);
        hipLaunchKernelGGL(k_grid_insert_both, dim3(nblk(nPrim)), dim3(BLOCK), 0, stream, nSF, d_SF.p, nSFE, d_SFE.p, x_dev, g, (int)nCells, infl, 1, capItems, stale,
            gridCount_.p, gridStart_.p, gridItems_.p);
        hipLaunchKernelGGL(k_narrow_pt, dim3(nblk(COOP * nSVI)), dim3(BLOCK), 0, stream, nSVI, d_SVI.p, d_SF.p, x_dev, pf, g, cellStartT, gridItems_.p,
            dHat, capPT, outPT_.p, counters_.p, capItems, stale, bucketCount_.p);
        hipLaunchKernelGGL(k_narrow_ee_cells, dim3((int)std::min<long long>(nblk(64 * nCells), 4096)), dim3(BLOCK), 0, stream, nSFE, d_SFE.p, x_dev, d_xRest.p, pf, g, (int)nCells, cellStartE,
            gridItems_.p, dHat, infl, capEE, outEE_.p, counters_.p + 1, capItems, stale, bucketCount_.p + nSVI);
        // the buckets of the record sort: needed only when the build stands, enqueued before the host knows (one scan; the GPU would idle through the read-back)
        scan(bucketCount_.p, bucketStart_.p, nB + 1);
        hipLaunchKernelGGL(k_publish_narrow, dim3(1), dim3(1), 0, stream, counters_.p, gridStart_.p + (nC2 - 1), box_dev, rbDev);
        HIP_CHECK(hipStreamSynchronize(stream));
        const int total = rb->cnt[3];
        for (int c = 0; c < 6; ++c) box_[c] = rb->box[c]; // the next grid
        if (rb->cnt[2]) continue; // the positions have left the old grid: nothing ran, once more on the fresh box
        if (total > capItems) { // the cell lists did not fit (the narrow phase walked truncated lists): grow and redo
            gridItems_.ensure((size_t)REC * ((size_t)total + (size_t)total / 4));
            bucketCount_.zero(stream);
            continue;
        }
        if (rb->cnt[0] > capPT || rb->cnt[1] > capEE) { // overflow: grow and redo
            capPT = std::max(capPT, rb->cnt[0] + rb->cnt[0] / 4);
            capEE = std::max(capEE, rb->cnt[1] + rb->cnt[1] / 4);
            bucketCount_.zero(stream);
            continue;
        }
        nPT = rb->cnt[0];
        nEE = rb->cnt[1];
        break;
    }
    // ---- sets assembled on the device (kernels above); one more read-back: the category totals and the number of merged tuples
    const int n = nPT + nEE;
    nCand_ = n;
    hostStale_ = true;
    d_csPTEE.ensure(2 * (size_t)std::max(n, 1));
    if (n == 0) {
        nActive_ = nPara_ = 0;
        countersDirty_ = false;
        return 0;
    }
    bucketSeg_.ensure(2 * (size_t)n);
    permPT_.ensure((size_t)n); // both permutations, the edge-edge one behind the point-triangle one
    flags_.ensure((size_t)n + 1);
    flagPos_.ensure((size_t)n + 1);
    // by (svI, sfI) and (eI, eJ) -- the order a serial scan emits --, the point-triangle records first
    hipLaunchKernelGGL(k_bucket_fill, dim3(nblk(n)), dim3(BLOCK), 0, stream, nPT, nEE, nSVI, outPT_.p, outEE_.p, bucketStart_.p, bucketCount_.p,
        reinterpret_cast<int2*>(bucketSeg_.p));
    hipLaunchKernelGGL(k_bucket_rank_classify, dim3(nblk(n + 1)), dim3(BLOCK), 0, stream, nPT, nEE, nSVI, nV, bucketStart_.p,
        reinterpret_cast<const int2*>(bucketSeg_.p), outPT_.p, outEE_.p, permPT_.p, d_csPTEE.p, flags_.p, dupCount_.p);
    const int* permPT = permPT_.p;
    const int* permEE = permPT_.p + nPT;
    scan(flags_.p, flagPos_.p, n + 1);
    scan(dupCount_.p, dupStart_.p, nV + 1);
    // Everything below runs without the host knowing the category totals (round 6: two synchronisations per build less): the sets are sized for the
    // worst case -- every candidate in one category --, the kernels read the totals where they need them (flagPos_[n]), and the counts come back at the end.
    const unsigned long long* totalsDev = flagPos_.p + n;
    d_active.ensure(4 * (size_t)n);
    d_para.ensure(4 * (size_t)n);
    d_paraEIEJ.ensure(2 * (size_t)n);
    dupTuple_.ensure(4 * (size_t)n);
    dupSorted_.ensure(4 * (size_t)n);
    hipLaunchKernelGGL(k_scatter_sets, dim3(nblk(n)), dim3(BLOCK), 0, stream, nPT, nEE, nSFE, outPT_.p, permPT, outEE_.p, permEE, flagPos_.p,
        d_active.p, dupTuple_.p, dupStart_.p, dupCount_.p, d_para.p, d_paraEIEJ.p, nV);
    // lexicographic order of (id0, id1, id2) = the map's: buckets by id0, each ordered by (id1, id2); one tuple per run, its multiplicity in the last slot
    hipLaunchKernelGGL(k_dup_rank, dim3(nblk(n)), dim3(BLOCK), 0, stream, totalsDev, nV, dupStart_.p, reinterpret_cast<const int4*>(dupTuple_.p),
        reinterpret_cast<int4*>(dupSorted_.p));
    hipLaunchKernelGGL(k_dup_runs, dim3(nblk(nV)), dim3(BLOCK), 0, stream, nV, dupStart_.p, reinterpret_cast<const int4*>(dupSorted_.p), runs_.p);
    scan(runs_.p, runPos_.p, nV + 1);
    hipLaunchKernelGGL(k_dup_emit, dim3(nblk(nV)), dim3(BLOCK), 0, stream, nV, totalsDev, dupStart_.p, reinterpret_cast<const int4*>(dupSorted_.p), runPos_.p,
        reinterpret_cast<int4*>(d_active.p), rbDev);
    HIP_CHECK(hipStreamSynchronize(stream));
    const unsigned long long totals = rb->totals;
    const int nDirect = (int)(totals & 0xffffffffull), nDup = (int)(totals >> 32), nPar = n - nDirect - nDup;
    const int nUnique = rb->nUnique;
    (void)nDup;
    countersDirty_ = false;
    nActive_ = nDirect + nUnique;
    nPara_ = nPar;
    return nActive_;
}

// host mirrors of the sets (tests, connectivity on a pattern change, friction lagging): fetched when somebody asks
void HipContact::syncHost() const
{
    if (!hostStale_) return;
    HipContact* self = const_cast<HipContact*>(this);
    std::vector<int> a(4 * (size_t)nActive_), p(4 * (size_t)nPara_), q(2 * (size_t)nPara_), c(2 * (size_t)nCand_);
    if (nActive_) d_active.download(a.data(), a.size(), stream);
    if (nPara_) {
        d_para.download(p.data(), p.size(), stream);
        d_paraEIEJ.download(q.data(), q.size(), stream);
    }
    if (nCand_) d_csPTEE.download(c.data(), c.size(), stream);
    HIP_CHECK(hipStreamSynchronize(stream));
    self->active.resize(nActive_);
    for (int i = 0; i < nActive_; ++i) self->active[i] = { a[4 * (size_t)i], a[4 * (size_t)i + 1], a[4 * (size_t)i + 2], a[4 * (size_t)i + 3] };
    self->para.resize(nPara_);
    self->paraEIEJ.resize(nPara_);
    for (int i = 0; i < nPara_; ++i) {
        self->para[i] = { p[4 * (size_t)i], p[4 * (size_t)i + 1], p[4 * (size_t)i + 2], p[4 * (size_t)i + 3] };
        self->paraEIEJ[i] = { q[2 * (size_t)i], q[2 * (size_t)i + 1] };
    }
    self->csPTEE.resize(nCand_);
    for (int i = 0; i < nCand_; ++i) self->csPTEE[i] = { c[2 * (size_t)i], c[2 * (size_t)i + 1] };
    self->hostStale_ = false;
}

bool HipContact::energyEnqueue(const double* x_dev, double dHat, double kappa, DevBuf<double>& partial, double* scalar_dev, const void* pubSrc, void* pubDst,
    int pubWords)
{
    if (nActive_ + nPara_ == 0) return false;
    int aB, aE, pB, pE; // this rank's share of the two lists
    shardRange(nActive_, aB, aE);
    shardRange(nPara_, pB, pE);
    const int n = (aE - aB) + (pE - pB);
    ContactView cv{ aE - aB, pE - pB, d_active.p + 4 * (size_t)aB, d_para.p + 4 * (size_t)pB, d_paraEIEJ.p + 2 * (size_t)pB, d_SFE.p, x_dev, d_xRest.p };
    const int nb = std::max(1, nblk(n));
    if (partial.n < (size_t)nb) partial.alloc(nb);
    if (n) hipLaunchKernelGGL(k_contact_energy, dim3(nb), dim3(BLOCK), 0, stream, cv, dHat, partial.p);
    const bool pub = pubWords > 0 && !(shardWorld > 1 && shardReduce); // (a sharded sum is complete only after the reduction below: the caller publishes)
    hipLaunchKernelGGL(k_reduce_scaled, dim3(1), dim3(BLOCK), 0, stream, partial.p, n ? nb : 0, kappa, scalar_dev, pub ? (const unsigned*)pubSrc : nullptr,
        pub ? (unsigned*)pubDst : nullptr, pub ? pubWords : 0);
    if (shardWorld > 1 && shardReduce) shardReduce(scalar_dev, 1);
    publishedByEnergy_ = pub;
    return true;
}
double HipContact::energy(const double* x_dev, double dHat, double kappa, DevBuf<double>& partial, double* scalar_dev)
{
    if (!energyEnqueue(x_dev, dHat, kappa, partial, scalar_dev)) return 0.0;
    readbackInit();
    hipLaunchKernelGGL(k_publish_u64, dim3(1), dim3(1), 0, stream, reinterpret_cast<const unsigned long long*>(scalar_dev), readback_.dev);
    HIP_CHECK(hipStreamSynchronize(stream));
    double out;
    std::memcpy(&out, readback_.p, sizeof(out));
    return out;
}

void HipContact::evaluateTuples(const double* x_dev, int n, const int* tuples4, double* val)
{
    if (n <= 0) return;
    tupleBuf_.upload(tuples4, 4 * (size_t)n, stream);
    tupleVal_.alloc((size_t)n);
    hipLaunchKernelGGL(k_evaluate_tuples, dim3(nblk(n)), dim3(BLOCK), 0, stream, n, tupleBuf_.p, x_dev, tupleVal_.p);
    tupleVal_.download(val, (size_t)n, stream);
}

void HipContact::jtMultiplyTuples(const double* x_dev, int nV, int n, const int* tuples4, const double* input, double coef, double* out_3nV)
{
    if (n <= 0) return;
    tupleBuf_.upload(tuples4, 4 * (size_t)n, stream);
    tupleVal_.upload(input, (size_t)n, stream);
    tupleOut_.upload(out_3nV, 3 * (size_t)nV, stream);
    hipLaunchKernelGGL(k_jt_tuples, dim3(nblk(n)), dim3(BLOCK), 0, stream, n, tupleBuf_.p, x_dev, tupleVal_.p, coef, tupleOut_.p);
    tupleOut_.download(out_3nV, 3 * (size_t)nV, stream);
}

void HipContact::gradientAdd(const double* x_dev, const int* dbc_dev, int nV, double dHat, double kappa, int projectDBC, double* grad_dev,
    bool useActive, bool usePara, const unsigned char* need_dev)
{
    // Always the WHOLE lists (round 4): which stencils a rank evaluates is decided by the caller's node mask (the optimizer's owner-computes plan), never by
    // an index range hidden in here -- ipcgpu_contact_gradient_add on a sharded context used to return a rank's share without saying so.
    const int nA = useActive ? nActive_ : 0, nP = usePara ? nPara_ : 0;
    const int n = nA + nP;
    if (n) {
        ContactView cv{ nA, nP, d_active.p, d_para.p, d_paraEIEJ.p, d_SFE.p, x_dev, d_xRest.p, need_dev };
        detBegin(8 * (size_t)n, 3, false, /*fillKeys=*/false, (size_t)nV); // the kernel writes every key
        hipLaunchKernelGGL(k_contact_gradient, dim3(nblk(n)), dim3(BLOCK), 0, stream, cv, dHat, kappa, GradSink{ detVals_.p, detKey_.p, detCount_.p });
        detReduce3(8 * (size_t)n, (size_t)nV, grad_dev);
    }
    hipLaunchKernelGGL(k_zero_projected, dim3(nblk(nV)), dim3(BLOCK), 0, stream, nV, dbc_dev, projectDBC, grad_dev);
}

void HipContact::hessianAdd(const double* x_dev, const int* dbc_dev, const HipLinSysSolver& lin, double dHat, double kappa, int projectDBC,
    double* a_dev, const unsigned char* need_dev, bool deferCheck)
{
    const int n = nActive_ + nPara_; // the whole lists; need_dev (or null) says which stencils this rank evaluates: see gradientAdd
    if (!n) return;
    ContactView cv{ nActive_, nPara_, d_active.p, d_para.p, d_paraEIEJ.p, d_SFE.p, x_dev, d_xRest.p, need_dev };
    CsrView m{ lin.d_ia.p, lin.d_ja.p };
    // counters: [0] pattern-miss flag, [1] Jacobi sweeps, [4 .. 11] the bin counts
    counters_.alloc(16);
    counters_.zero(stream);
    hessPerm_.ensure((size_t)NBINS * (size_t)n);
    hipLaunchKernelGGL(k_bin_stencils, dim3(nblk(n)), dim3(BLOCK), 0, stream, cv, counters_.p + 4, hessPerm_.p, n);
    const size_t nSlots = (size_t)HSLOTS * (size_t)n;
    const size_t nKeys = lin.ja.size() / 3 + 1; // buckets of the block positions
    detBegin(nSlots, 9, true, /*fillKeys=*/false, nKeys); // the kernel writes every key
    const HessBins bins{ counters_.p + 4, hessPerm_.p, n };
    hipLaunchKernelGGL(k_contact_hessian, dim3(nblk(n, HESS_W) + NBINS), dim3(HESS_W), 0, stream, cv, bins, m, dbc_dev, projectDBC, dHat, kappa,
        BlockSink{ detVals_.p, detKey_.p, detRow_.p, detCount_.p }, counters_.p);
    detReduceBlocks(nSlots, nKeys, lin.d_ia.p, a_dev);
    if (deferCheck) { // the flag goes to mapped host memory behind the pass; the caller looks at it after its next synchronisation (takeHessianError)
        readbackInit();
        hipLaunchKernelGGL(k_publish_int, dim3(1), dim3(1), 0, stream, (const int*)counters_.p, reinterpret_cast<int*>(readback_.dev + 15));
        hessErrPending_ = true;
        return;
    }
    int err[2];
    counters_.download(err, 2, stream);
    if (std::getenv("IPCGPU_DEBUG")) std::fprintf(stderr, "[ipcgpu] barrier Hessian: %d stencils, %.2f Jacobi sweeps on average\n", n, (double)err[1] / n);
    if (err[0]) throw StateError("barrier Hessian touches a node pair outside the CSR pattern: call set_pattern with the contact connectivity first");
}

void HipContact::takeHessianError()
{
    if (!hessErrPending_) return;
    hessErrPending_ = false;
    if (*reinterpret_cast<const volatile int*>(readback_.p + 15))
        throw StateError("barrier Hessian touches a node pair outside the CSR pattern: call set_pattern with the contact connectivity first");
}

bool HipContact::patternCovers(const HipLinSysSolver& lin)
{
    const int n = nActive_ + nPara_;
    if (!n) return true;
    ContactView cv{ nActive_, nPara_, d_active.p, d_para.p, d_paraEIEJ.p, d_SFE.p, nullptr, d_xRest.p };
    CsrView m{ lin.d_ia.p, lin.d_ja.p };
    counters_.alloc(16);
    counters_.zero(stream);
    hipLaunchKernelGGL(k_pattern_check, dim3(nblk(n)), dim3(BLOCK), 0, stream, cv, m, counters_.p);
    int miss = 0;
    counters_.download(&miss, 1, stream);
    return miss == 0;
}

// ---- lagged friction: host side -------------------------------------------------------------------------------------
void HipContact::frictionLagClear() { fricSet.clear(); }

void HipContact::frictionLagUpdate(const double* x_dev, double dHat, double kappa)
{
    syncHost();
    fricSet = active; // MMActiveSet_lastH = MMActiveSet (Optimizer.cpp:1596-1598); d_active holds the same tuples
    const int n = (int)fricSet.size();
    if (!n) return;
    d_fricSet.ensure(4 * (size_t)n);
    HIP_CHECK(hipMemcpyAsync(d_fricSet.p, d_active.p, 4 * (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, stream));
    d_fricLambda.ensure(n);
    d_fricCoord.ensure(2 * (size_t)n);
    d_fricBasis.ensure(6 * (size_t)n);
    const bool scaled = hasObstacle && (fricScaleSelf != 1.0 || fricScaleObst != 1.0);
    hipLaunchKernelGGL(k_friction_lag, dim3(nblk(n)), dim3(BLOCK), 0, stream, n, d_fricSet.p, x_dev, dHat, kappa, scaled ? (const int*)d_obst.p : nullptr,
        fricScaleSelf, fricScaleObst, d_fricLambda.p, d_fricCoord.p, d_fricBasis.p);
}

void HipContact::frictionGet(double* lambda, double* coord2, double* basis6)
{
    const size_t n = fricSet.size();
    if (!n) return;
    d_fricLambda.download(lambda, n, stream);
    d_fricCoord.download(coord2, 2 * n, stream);
    d_fricBasis.download(basis6, 6 * n, stream);
}

double HipContact::frictionEnergy(const double* x_dev, const double* xt_dev, double eps2, double coef, DevBuf<double>& partial, double* scalar_dev)
{
    const int n = (int)fricSet.size();
    if (!n) return 0.0;
    FrictionView fv{ n, d_fricSet.p, d_fricLambda.p, d_fricCoord.p, d_fricBasis.p };
    const int nb = nblk(n);
    if (partial.n < (size_t)nb) partial.alloc(nb);
    hipLaunchKernelGGL(k_friction_energy, dim3(nb), dim3(BLOCK), 0, stream, fv, x_dev, xt_dev, eps2, partial.p);
    hipLaunchKernelGGL(k_reduce_scaled, dim3(1), dim3(BLOCK), 0, stream, partial.p, nb, coef, scalar_dev, (const unsigned*)nullptr, (unsigned*)nullptr, 0);
    double out = 0.0;
    HIP_CHECK(hipMemcpyAsync(&out, scalar_dev, sizeof(double), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    return out;
}

// ---- deterministic scatter, host side: slots + keys + bucket counters for one launch, then scan, fill and sum the buckets --------------------
void HipContact::detBegin(size_t nSlots, int valsPerSlot, bool withRow, bool fillKeys, size_t nKeys)
{
    if (nSlots > (size_t)INT_MAX || nKeys + 1 > (size_t)INT_MAX) throw StateError("too many contact contributions for one scatter pass");
    detVals_.ensure(nSlots * (size_t)valsPerSlot);
    detKey_.ensure(nSlots);
    detSeg_.ensure(nSlots);
    detSorted_.ensure(nSlots);
    if (withRow) detRow_.ensure(nSlots);
    if (detCount_.n < nKeys + 1 || detDirty_) { // the counters return to zero with every pass that runs to its end (k_det_fill)
        detCount_.ensure(nKeys + 1);
        detCount_.zero(stream);
    }
    detStart_.ensure(nKeys + 1);
    detDirty_ = true;
    if (fillKeys) HIP_CHECK(hipMemsetAsync(detKey_.p, 0xFF, nSlots * sizeof(unsigned), stream)); // KEY_NONE: slots the kernel does not write
}
void HipContact::detBuckets(size_t nSlots, size_t nKeys, int div)
{
    size_t bytes = 0;
    hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, detCount_.p, detStart_.p, (int)nKeys + 1, stream);
    if (scanTmp_.n < bytes) scanTmp_.alloc(bytes + bytes / 4);
    hipcub::DeviceScan::ExclusiveSum((void*)scanTmp_.p, bytes, detCount_.p, detStart_.p, (int)nKeys + 1, stream);
    hipLaunchKernelGGL(k_det_fill, dim3(nblk((int)nSlots)), dim3(BLOCK), 0, stream, (int)nSlots, detKey_.p, div, detStart_.p, detCount_.p, detSeg_.p);
    hipLaunchKernelGGL(k_det_rank, dim3(nblk((int)nSlots)), dim3(BLOCK), 0, stream, (int)nSlots, detKey_.p, div, detStart_.p, (int)nKeys, detSeg_.p, detSorted_.p);
    detDirty_ = false;
}
void HipContact::detReduce3(size_t nSlots, size_t nKeys, double* grad_dev)
{
    detBuckets(nSlots, nKeys, 1);
    hipLaunchKernelGGL(k_bucket_sum3, dim3(nblk((int)nKeys)), dim3(BLOCK), 0, stream, (int)nKeys, detStart_.p, detSorted_.p, detVals_.p, grad_dev);
}
void HipContact::detReduceBlocks(size_t nSlots, size_t nKeys, const int* ia_dev, double* a_dev)
{
    detBuckets(nSlots, nKeys, 3);
    hipLaunchKernelGGL(k_bucket_sum_blocks, dim3(nblk((int)nKeys)), dim3(BLOCK), 0, stream, (int)nKeys, detStart_.p, detSorted_.p, detKey_.p, detVals_.p, detRow_.p,
        ia_dev, a_dev);
}

void HipContact::frictionGradientAdd(const double* x_dev, const double* xt_dev, double eps2, double coef, double* grad_dev)
{
    const int n = (int)fricSet.size();
    if (!n) return;
    FrictionView fv{ n, d_fricSet.p, d_fricLambda.p, d_fricCoord.p, d_fricBasis.p };
    detBegin(8 * (size_t)n, 3, false, true, d_xRest.n / 3); // (d_xRest: three rest coordinates per node)
    hipLaunchKernelGGL(k_friction_gradient, dim3(nblk(n)), dim3(BLOCK), 0, stream, fv, x_dev, xt_dev, eps2, coef, GradSink{ detVals_.p, detKey_.p, detCount_.p });
    detReduce3(8 * (size_t)n, d_xRest.n / 3, grad_dev);
}

void HipContact::frictionHessianAdd(const double* x_dev, const double* xt_dev, const int* dbc_dev, const HipLinSysSolver& lin, double eps2, double coef,
    int projectDBC, double* a_dev)
{
    const int n = (int)fricSet.size();
    if (!n) return;
    FrictionView fv{ n, d_fricSet.p, d_fricLambda.p, d_fricCoord.p, d_fricBasis.p };
    CsrView m{ lin.d_ia.p, lin.d_ja.p };
    counters_.alloc(16);
    counters_.zero(stream);
    const size_t nKeys = lin.ja.size() / 3 + 1;
    detBegin(16 * (size_t)n, 9, true, true, nKeys);
    hipLaunchKernelGGL(k_friction_hessian, dim3(nblk(n)), dim3(BLOCK), 0, stream, fv, m, x_dev, xt_dev, dbc_dev, projectDBC, eps2, coef,
        BlockSink{ detVals_.p, detKey_.p, detRow_.p, detCount_.p }, counters_.p);
    detReduceBlocks(16 * (size_t)n, nKeys, lin.d_ia.p, a_dev);
    int err[2];
    counters_.download(err, 2, stream);
    if (err[0]) throw StateError("friction Hessian touches a node pair outside the CSR pattern: the pattern must contain the lagged set's connectivity");
}

void HipContact::frictionConnectivity(std::vector<std::pair<int, int>>& pairs) const
{
    auto link = [&](int a, int b) {
        if (a != b) pairs.push_back({ std::min(a, b), std::max(a, b) });
    };
    for (const auto& c : fricSet) { // SelfCollisionHandler.cpp:330-376 on MMActiveSet_lastH (Optimizer.cpp:3565-3566)
        if (c[0] >= 0) {
            link(c[0], c[2]);
            link(c[0], c[3]);
            link(c[1], c[2]);
            link(c[1], c[3]);
        }
        else {
            const int v0 = -c[0] - 1;
            link(v0, c[1]);
            if (c[2] >= 0) link(v0, c[2]);
            if (c[2] >= 0 && c[3] >= 0) link(v0, c[3]);
        }
    }
}

// Every node pair a candidate primitive pair can ever couple, whatever its closest-feature type: vertex x the three triangle
// nodes, the 2 x 2 end points of an edge pair.  A superset of connectivity() for the same set; used for the look-ahead pattern
// (a stencil that slides from point-point to point-triangle needs no new matrix blocks then).
// node pairs of the candidate list as 64-bit keys (lo << 32 | hi), four per candidate (a point-triangle candidate has three: the
// fourth slot, like a pair of equal nodes, is the all-ones sentinel that sorts last)
__global__ void k_cand_pair_keys(int n, const int* __restrict__ cs, const int* __restrict__ SVI, const int* __restrict__ SF,
    const int* __restrict__ SFE, unsigned long long* __restrict__ key)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c0 = cs[2 * (size_t)i], c1 = cs[2 * (size_t)i + 1];
    auto mk = [](int a, int b) { return a == b ? ~0ull : (((unsigned long long)(unsigned)min(a, b) << 32) | (unsigned)max(a, b)); };
    unsigned long long* k = key + 4 * (size_t)i;
    if (c0 < 0) {
        const int v = SVI[-c0 - 1];
        for (int q = 0; q < 3; ++q) k[q] = mk(v, SF[3 * (size_t)c1 + q]);
        k[3] = ~0ull;
    }
    else {
        const int a0 = SFE[2 * (size_t)c0], a1 = SFE[2 * (size_t)c0 + 1], b0 = SFE[2 * (size_t)c1], b1 = SFE[2 * (size_t)c1 + 1];
        k[0] = mk(a0, b0);
        k[1] = mk(a0, b1);
        k[2] = mk(a1, b0);
        k[3] = mk(a1, b1);
    }
}

// The same pairs as candidateConnectivity, sorted and without duplicates, formed on the device: radix sort + unique of the keys
// (a few hundred thousand pairs: 20 ms of std::sort on the host, per pattern change, twice).  Appends to `pairs`.
void HipContact::candidateConnectivitySorted(std::vector<std::pair<int, int>>& pairs)
{
    const int n = nCand_;
    if (n <= 0) return;
    const size_t m = 4 * (size_t)n;
    sortKeyIn_.ensure(m + 1);
    sortKeyOut_.ensure(m + 1);
    hipLaunchKernelGGL(k_cand_pair_keys, dim3(nblk(n)), dim3(BLOCK), 0, stream, n, d_csPTEE.p, d_SVI.p, d_SF.p, d_SFE.p, sortKeyIn_.p);
    auto tmp = [&](size_t bytes) {
        if (scanTmp_.n < bytes) scanTmp_.alloc(bytes + bytes / 4);
        return (void*)scanTmp_.p;
    };
    {
        size_t bytes = 0;
        hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, sortKeyIn_.p, sortKeyOut_.p, (int)m, 0, 64, stream);
        hipcub::DeviceRadixSort::SortKeys(tmp(bytes), bytes, sortKeyIn_.p, sortKeyOut_.p, (int)m, 0, 64, stream);
    }
    unsigned long long* d_cnt = sortKeyOut_.p + m; // one spare slot behind the keys
    {
        size_t bytes = 0;
        hipcub::DeviceSelect::Unique(nullptr, bytes, sortKeyOut_.p, sortKeyIn_.p, reinterpret_cast<int*>(d_cnt), (int)m, stream);
        hipcub::DeviceSelect::Unique(tmp(bytes), bytes, sortKeyOut_.p, sortKeyIn_.p, reinterpret_cast<int*>(d_cnt), (int)m, stream);
    }
    int cnt = 0;
    HIP_CHECK(hipMemcpyAsync(&cnt, d_cnt, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    std::vector<unsigned long long> keys((size_t)cnt);
    if (cnt) HIP_CHECK(hipMemcpyAsync(keys.data(), sortKeyIn_.p, (size_t)cnt * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    pairs.reserve(pairs.size() + keys.size());
    for (unsigned long long k : keys)
        if (k != ~0ull) pairs.push_back({ (int)(k >> 32), (int)(k & 0xffffffffull) });
}

void HipContact::candidateConnectivity(std::vector<std::pair<int, int>>& pairs) const
{
    syncHost();
    auto link = [&](int a, int b) {
        if (a != b) pairs.push_back({ std::min(a, b), std::max(a, b) });
    };
    for (const auto& c : csPTEE) {
        if (c[0] < 0) {
            const int v = SVI[-c[0] - 1];
            for (int k = 0; k < 3; ++k) link(v, SF[c[1] + (size_t)nSF * k]);
        }
        else {
            const auto &ea = SFEdges[c[0]], &eb = SFEdges[c[1]];
            link(ea.first, eb.first);
            link(ea.first, eb.second);
            link(ea.second, eb.first);
            link(ea.second, eb.second);
        }
    }
}

void HipContact::connectivity(std::vector<std::pair<int, int>>& pairs) const
{
    syncHost();
    pairs.clear();
    auto link = [&](int a, int b) {
        if (a != b) pairs.push_back({ std::min(a, b), std::max(a, b) });
    };
    auto nodesOf = [&](const std::array<int, 4>& c, int* node, int& n, bool& ee) {
        ee = c[0] >= 0;
        if (ee) {
            n = 4;
            for (int k = 0; k < 4; ++k) node[k] = c[k];
        }
        else {
            node[0] = -c[0] - 1;
            node[1] = c[1];
            n = 2;
            if (c[2] >= 0) node[n++] = c[2];
            if (c[2] >= 0 && c[3] >= 0) node[n++] = c[3];
        }
    };
    for (const auto& c : active) { // SelfCollisionHandler.cpp:330-376
        int node[4], n;
        bool ee;
        nodesOf(c, node, n, ee);
        if (ee) {
            link(node[0], node[2]);
            link(node[0], node[3]);
            link(node[1], node[2]);
            link(node[1], node[3]);
        }
        else
            for (int k = 1; k < n; ++k) link(node[0], node[k]);
    }
    for (size_t i = 0; i < para.size(); ++i) { // :378-415
        int en[4];
        if (para[i][3] >= 0)
            for (int k = 0; k < 4; ++k) en[k] = para[i][k];
        else {
            en[0] = SFEdges[paraEIEJ[i][0]].first;
            en[1] = SFEdges[paraEIEJ[i][0]].second;
            en[2] = SFEdges[paraEIEJ[i][1]].first;
            en[3] = SFEdges[paraEIEJ[i][1]].second;
        }
        link(en[0], en[2]);
        link(en[0], en[3]);
        link(en[1], en[2]);
        link(en[1], en[3]);
    }
    std::sort(pairs.begin(), pairs.end());
    pairs.erase(std::unique(pairs.begin(), pairs.end()), pairs.end());
}

namespace {
__global__ void k_max_speed(int n, const int* __restrict__ ids, const double* __restrict__ p, unsigned long long* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0.0;
    if (i < n) {
        const int v = ids ? ids[i] : i;
        s = sqrt(p[3 * (size_t)v] * p[3 * (size_t)v] + p[3 * (size_t)v + 1] * p[3 * (size_t)v + 1] + p[3 * (size_t)v + 2] * p[3 * (size_t)v + 2]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s = fmax(s, __shfl_down(s, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(s));
}
} // namespace

double HipContact::maxSurfaceSpeed(const double* p_dev)
{
    ccdOut_.alloc(4);
    ccdOut_.zero(stream);
    if (nSVI) hipLaunchKernelGGL(k_max_speed, dim3(nblk(nSVI)), dim3(BLOCK), 0, stream, nSVI, d_SVI.p, p_dev, ccdOut_.p);
    unsigned long long bits = 0;
    HIP_CHECK(hipMemcpyAsync(&bits, ccdOut_.p, sizeof(bits), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    double v;
    std::memcpy(&v, &bits, sizeof(v));
    return v;
}

HipContact::GridHost HipContact::makeGrid(const HipMesh& mesh, const double* x_dev, const double* p_dev, double alpha, double minCell)
{
    const int nV = mesh.nV;
    const int nb = nblk(nV);
    bboxPartial_.ensure(6 * (size_t)nb);
    hipLaunchKernelGGL(k_bbox_partial, dim3(nb), dim3(BLOCK), 0, stream, nV, x_dev, bboxPartial_.p);
    std::vector<double> part(6 * (size_t)nb);
    bboxPartial_.download(part.data(), part.size(), stream);
    GridHost g;
    double hi[3];
    for (int c = 0; c < 3; ++c) {
        g.lo[c] = 1e300;
        hi[c] = -1e300;
    }
    for (int b = 0; b < nb; ++b)
        for (int c = 0; c < 3; ++c) {
            g.lo[c] = std::min(g.lo[c], part[6 * (size_t)b + c]);
            hi[c] = std::max(hi[c], part[6 * (size_t)b + 3 + c]);
        }
    double reach = 0.0;
    if (p_dev) { // swept boxes reach at most alpha * max |p| beyond the current bounding box
        ccdOut_.alloc(4);
        ccdOut_.zero(stream);
        hipLaunchKernelGGL(k_max_speed, dim3(nblk(nV)), dim3(BLOCK), 0, stream, nV, (const int*)nullptr, p_dev, ccdOut_.p);
        unsigned long long bits = 0;
        HIP_CHECK(hipMemcpyAsync(&bits, ccdOut_.p, sizeof(bits), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        double pm;
        std::memcpy(&pm, &bits, sizeof(pm));
        reach = alpha * pm;
    }
    for (int c = 0; c < 3; ++c) {
        g.lo[c] -= reach;
        hi[c] += reach;
    }
    g.h = std::max(minCell, reach);
    for (;;) {
        g.nCells = 1;
        for (int c = 0; c < 3; ++c) {
            g.dim[c] = std::max(1, (int)std::floor((hi[c] - g.lo[c]) / g.h) + 1);
            g.nCells *= g.dim[c];
        }
        if (g.nCells <= (1LL << 26)) break;
        g.h *= 1.5;
    }
    return g;
}

void HipContact::buildCells(const GridHost& gh, int nPrim, int nv, const int* prim, const double* x_dev, const double* p_dev, double alpha, double infl,
    DevBuf<int>& cnt, DevBuf<int>& start, DevBuf<int>& items)
{
    Grid g;
    for (int c = 0; c < 3; ++c) {
        g.lo[c] = gh.lo[c];
        g.dim[c] = gh.dim[c];
    }
    g.h = gh.h;
    const long long nCells = gh.nCells;
    cnt.ensure((size_t)nCells + 1);
    start.ensure((size_t)nCells + 1);
    auto insert = [&](int mode) {
        cnt.zeroN((size_t)nCells + 1, stream);
        if (p_dev)
            hipLaunchKernelGGL(k_grid_insert_swept, dim3(nblk(nPrim)), dim3(BLOCK), 0, stream, nPrim, nv, prim, x_dev, p_dev, alpha, g, mode, cnt.p,
                start.p, items.p);
        else
            hipLaunchKernelGGL(k_grid_insert, dim3(nblk(nPrim)), dim3(BLOCK), 0, stream, nPrim, nv == 3 ? 1 : 0, prim, x_dev, g, infl, mode, cnt.p, start.p,
                items.p);
    };
    insert(0);
    size_t tmpBytes = 0;
    HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmpBytes, cnt.p, start.p, (int)nCells + 1, stream));
    if (scanTmp_.n < tmpBytes) scanTmp_.alloc(tmpBytes);
    HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(scanTmp_.p, tmpBytes, cnt.p, start.p, (int)nCells + 1, stream));
    int total = 0;
    HIP_CHECK(hipMemcpyAsync(&total, start.p + nCells, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    items.ensure((size_t)(p_dev ? 1 : REC) * std::max(1, total)); // k_grid_insert writes records, k_grid_insert_swept indices
    insert(1);
}

static void decodeCcdOut(const unsigned long long* h, double stepSize, double* outStep, int* pair2)
{
    double t;
    std::memcpy(&t, &h[0], sizeof(t));
    if (h[1] == ~0ull || !(t < stepSize)) { // no pair limits the step
        *outStep = stepSize;
        if (pair2) pair2[0] = pair2[1] = 0;
        return;
    }
    *outStep = t;
    if (pair2) {
        const int isEE = (int)(h[1] >> 62), i = (int)((h[1] >> 31) & 0x7FFFFFFF), j = (int)(h[1] & 0x7FFFFFFF);
        pair2[0] = isEE ? i : -i - 1;
        pair2[1] = j;
    }
}

double HipContact::ccdPartial(const double* x_dev, const double* p_dev, double slackness, double stepSize, int* pair2)
{
    const int n = nCand_;
    if (pair2) pair2[0] = pair2[1] = 0;
    if (!n) return stepSize;
    ccdOut_.alloc(4);
    const unsigned long long init[2] = { ~0ull, ~0ull };
    HIP_CHECK(hipMemcpyAsync(ccdOut_.p, init, sizeof(init), hipMemcpyHostToDevice, stream));
    CcdOut o{ ccdOut_.p, ccdOut_.p + 1 };
    for (int pass = 0; pass < 2; ++pass)
        hipLaunchKernelGGL(k_ccd_list, dim3(nblk(n)), dim3(BLOCK), 0, stream, n, d_csPTEE.p, d_SVI.p, d_SF.p, d_SFE.p, x_dev, p_dev, slackness, stepSize, pass,
            o, (const unsigned long long*)nullptr);
    unsigned long long h[2];
    HIP_CHECK(hipMemcpyAsync(h, ccdOut_.p, sizeof(h), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    double out;
    decodeCcdOut(h, stepSize, &out, pair2);
    return out;
}

// The step bounds a Newton iteration of a contact scene asks for one after the other (Optimizer.cpp:1884-1953): the inversion filter's root (already on the
// device, *filterDev, applied by the filter's rule -- Energy.cpp:565-581: only when 0 < root < step), the partial CCD over the candidates bounded by the
// result, and the largest surface speed for the CFL test -- enqueued together and read back with ONE synchronisation (three before round 6).
void HipContact::stepBounds(const double* x_dev, const double* p_dev, double slackness, double stepSize, const double* filterDev, double* alphaOut, int* pair2,
    double* pMaxOut)
{
    ccdOut_.alloc(4);
    readbackInit();
    hipLaunchKernelGGL(k_step_bounds_init, dim3(1), dim3(1), 0, stream, stepSize, filterDev, ccdOut_.p);
    const int n = nCand_;
    CcdOut o{ ccdOut_.p, ccdOut_.p + 1 };
    for (int pass = 0; pass < 2 && n; ++pass)
        hipLaunchKernelGGL(k_ccd_list, dim3(nblk(n)), dim3(BLOCK), 0, stream, n, d_csPTEE.p, d_SVI.p, d_SF.p, d_SFE.p, x_dev, p_dev, slackness, stepSize, pass,
            o, (const unsigned long long*)(ccdOut_.p + 3));
    if (nSVI) hipLaunchKernelGGL(k_max_speed, dim3(nblk(nSVI)), dim3(BLOCK), 0, stream, nSVI, d_SVI.p, p_dev, ccdOut_.p + 2);
    hipLaunchKernelGGL(k_publish_u64x4, dim3(1), dim3(4), 0, stream, (const unsigned long long*)ccdOut_.p, readback_.dev);
    HIP_CHECK(hipStreamSynchronize(stream));
    const unsigned long long* h = readback_.p;
    double filtered;
    std::memcpy(&filtered, &h[3], sizeof(filtered));
    std::memcpy(pMaxOut, &h[2], sizeof(double));
    decodeCcdOut(h, filtered, alphaOut, pair2);
}

double HipContact::ccdFull(const HipMesh& mesh, const double* x_dev, const double* p_dev, const int* dbc_dev, double slackness, double stepSize,
    int* pair2, int* nCand)
{
    if (!surfaceSet) throw StateError("ccd before set_surface");
    const int* pf = pairFlags(mesh.nV, dbc_dev);
    const GridHost gh = makeGrid(mesh, x_dev, p_dev, stepSize, mesh.avgEdgeLen);
    buildCells(gh, nSF, 3, d_SF.p, x_dev, p_dev, stepSize, 0.0, cellCountT_, cellStartT_, cellItemsT_);
    buildCells(gh, nSFE, 2, d_SFE.p, x_dev, p_dev, stepSize, 0.0, cellCountE_, cellStartE_, cellItemsE_);
    Grid g;
    for (int c = 0; c < 3; ++c) {
        g.lo[c] = gh.lo[c];
        g.dim[c] = gh.dim[c];
    }
    g.h = gh.h;
    ccdOut_.alloc(4);
    const unsigned long long init[2] = { ~0ull, ~0ull };
    HIP_CHECK(hipMemcpyAsync(ccdOut_.p, init, sizeof(init), hipMemcpyHostToDevice, stream));
    counters_.alloc(16);
    counters_.zero(stream);
    CcdOut o{ ccdOut_.p, ccdOut_.p + 1 };
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(k_ccd_full_pt, dim3(nblk(nSVI)), dim3(BLOCK), 0, stream, nSVI, d_SVI.p, d_SF.p, x_dev, p_dev, pf, g, cellStartT_.p,
            cellItemsT_.p, stepSize, slackness, pass, o, counters_.p);
        hipLaunchKernelGGL(k_ccd_full_ee, dim3(nblk(nSFE)), dim3(BLOCK), 0, stream, nSFE, d_SFE.p, x_dev, p_dev, pf, g, cellStartE_.p, cellItemsE_.p,
            stepSize, slackness, pass, o, counters_.p);
    }
    unsigned long long h[2];
    int cnt[2];
    HIP_CHECK(hipMemcpyAsync(h, ccdOut_.p, sizeof(h), hipMemcpyDeviceToHost, stream));
    counters_.download(cnt, 2, stream);
    if (nCand) *nCand = cnt[0];
    double out;
    decodeCcdOut(h, stepSize, &out, pair2);
    return out;
}

double HipContact::ccdFullReference(const HipMesh& mesh, const double* x_dev, const double* p_dev, const int* dbc_dev, double slackness, double alpha,
    double* alphaCapped, int* arg3, int* nCand)
{
    if (!surfaceSet) throw StateError("ccd before set_surface");
    if (arg3) arg3[0] = arg3[1] = arg3[2] = -1;
    if (nCand) *nCand = 0;
    if (alphaCapped) *alphaCapped = alpha;
    if (!nSVI) return alpha;
    const int* pf = pairFlags(mesh.nV, dbc_dev);
    // the cap (SpatialHash.hpp:603-618): mean |component| of p over the surface nodes against the cell size; corner and extent (:620-634): all nodes now, the
    // surface nodes at the capped step -- all on the device, ONE read-back (k_ref_cap, k_ref_bbox_swept_dev, k_ref_box_final)
    const int nbS = nblk(nSVI), nbV = nblk(mesh.nV);
    const size_t offAbs = 0, offBoxV = 2 * (size_t)nbS, offBoxS = offBoxV + 6 * (size_t)nbV, offOut = offBoxS + 6 * (size_t)nbS;
    bboxPartial_.ensure(offOut + 8);
    double* outDev = bboxPartial_.p + offOut;
    const double voxelSize = mesh.avgEdgeLen / 3.0;
    readbackInit();
    hipLaunchKernelGGL(k_ref_abs_sum, dim3(nbS), dim3(BLOCK), 0, stream, nSVI, d_SVI.p, hasObstacle ? (const int*)d_obst.p : (const int*)nullptr, p_dev,
        bboxPartial_.p + offAbs);
    hipLaunchKernelGGL(k_ref_cap, dim3(1), dim3(64), 0, stream, nbS, bboxPartial_.p + offAbs, alpha, voxelSize, outDev);
    hipLaunchKernelGGL(k_bbox_partial, dim3(nbV), dim3(BLOCK), 0, stream, mesh.nV, x_dev, bboxPartial_.p + offBoxV);
    hipLaunchKernelGGL(k_ref_bbox_swept_dev, dim3(nbS), dim3(BLOCK), 0, stream, nSVI, d_SVI.p, x_dev, p_dev, (const double*)outDev, bboxPartial_.p + offBoxS);
    hipLaunchKernelGGL(k_ref_box_final, dim3(1), dim3(BLOCK), 0, stream, nbV, bboxPartial_.p + offBoxV, nbS, bboxPartial_.p + offBoxS, outDev,
        reinterpret_cast<double*>(readback_.dev));
    HIP_CHECK(hipStreamSynchronize(stream));
    const double* hb = reinterpret_cast<const double*>(readback_.p);
    alpha = hb[0];
    if (alphaCapped) *alphaCapped = alpha;
    double lb[3] = { hb[1], hb[2], hb[3] }, rt[3] = { hb[4], hb[5], hb[6] };
    RefGrid g;
    g.oneDiv = 1.0 / voxelSize;
    {
        int minCount = 1 << 30;
        double maxRange = 0.0;
        for (int c = 0; c < 3; ++c) {
            minCount = std::min(minCount, (int)std::ceil((rt[c] - lb[c]) * g.oneDiv));
            maxRange = std::max(maxRange, rt[c] - lb[c]);
        }
        if (minCount <= 0) g.oneDiv = 1.0 / (maxRange * 1.01);
    }
    int nRef[3];
    for (int c = 0; c < 3; ++c) {
        g.lb[c] = lb[c];
        nRef[c] = std::max(1, (int)std::floor((rt[c] - lb[c]) * g.oneDiv) + 1);
    }
    long long nCells;
#ifndef REF_GRID_MIN_M
#define REF_GRID_MIN_M 2 // search cells of 2 x 2 x 2 reference voxels (1 until round 6: profiles/r06_ref_grid_coarsening_ab.txt); pairs are accepted on the fine voxel indices either way
#endif
    for (g.m = REF_GRID_MIN_M;; ++g.m) {
        nCells = 1;
        for (int c = 0; c < 3; ++c) {
            g.dim[c] = (nRef[c] + g.m - 1) / g.m;
            nCells *= g.dim[c];
        }
        if (nCells <= (1LL << 23)) break;
    }
    refVbox_.ensure(6 * (size_t)nSVI);
    hipLaunchKernelGGL(k_ref_vbox, dim3(nbS), dim3(BLOCK), 0, stream, nSVI, d_SVI.p, x_dev, p_dev, alpha, g, refVbox_.p);
    // The three cell structures (vertices, edges, triangles) in ONE counter array of 3 (nCells + 1) entries, one scan, one item list (round 6; three times
    // clear - count - scan - read back - clear - fill before: the six clears of up to 32 MB each were ~0.3 ms of host time apiece, the whole sweep 3.3 ms of
    // which the kernels took 0.9): the last entry of each third stays zero, the scan makes the offsets of the later thirds absolute, the fill counts the
    // counters back down (k_ref_insert), and the host waits once, for the total.
    const size_t nC1 = (size_t)nCells + 1;
    if (refCount_.n < 3 * nC1 || refDirty_) { // (dirty: a sweep that was left between its count and its fill)
        refCount_.ensure(3 * nC1);
        refCount_.zero(stream);
    }
    refDirty_ = true;
    refStart_.ensure(3 * nC1);
    const int nPrims[3] = { nSVI, nSFE, nSF }, nvs[3] = { 1, 2, 3 };
    const int* prims[3] = { d_SVI.p, d_SFE.p, d_SF.p };
    for (int q = 0; q < 3; ++q)
        if (nPrims[q])
            hipLaunchKernelGGL(k_ref_insert, dim3(nblk(nPrims[q])), dim3(BLOCK), 0, stream, nPrims[q], nvs[q], prims[q], d_v2sv.p, refVbox_.p, g, 0,
                refCount_.p + q * nC1, (const int*)nullptr, (int*)nullptr);
    {
        size_t tmpBytes = 0;
        HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmpBytes, refCount_.p, refStart_.p, (int)(3 * nC1), stream));
        if (scanTmp_.n < tmpBytes) scanTmp_.alloc(tmpBytes + tmpBytes / 4);
        HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(scanTmp_.p, tmpBytes, refCount_.p, refStart_.p, (int)(3 * nC1), stream));
    }
    readbackInit();
    hipLaunchKernelGGL(k_publish_int, dim3(1), dim3(1), 0, stream, (const int*)(refStart_.p + 3 * nC1 - 1), reinterpret_cast<int*>(readback_.dev));
    HIP_CHECK(hipStreamSynchronize(stream));
    const int totalItems = *reinterpret_cast<const int*>(readback_.p);
    refItems_.ensure((size_t)REC * std::max(1, totalItems));
    for (int q = 0; q < 3; ++q)
        if (nPrims[q])
            hipLaunchKernelGGL(k_ref_insert, dim3(nblk(nPrims[q])), dim3(BLOCK), 0, stream, nPrims[q], nvs[q], prims[q], d_v2sv.p, refVbox_.p, g, 1,
                refCount_.p + q * nC1, (const int*)(refStart_.p + q * nC1), refItems_.p);
    refDirty_ = false;
    ccdOut_.alloc(4);
    counters_.alloc(16);
    // counters_: [0] queried pairs, [1] pairs that returned a time inside the step (the hit list)
    constexpr int HIT_CAP = 1 << 20;
    const bool twoPass = false; // (true: the limiting pair by a second run of the sweep, rounds 1-2; profiles/r03m_contact_bench_lds_jacobi_two_pass_ccd.json)
    if (!twoPass) ccdHits_.ensure(2 * (size_t)HIT_CAP);
    HIP_CHECK(hipMemsetAsync(ccdOut_.p, 0xFF, 3 * sizeof(unsigned long long), stream)); // "no time yet" = all ones
    counters_.zero(stream);
    CcdOut o{ ccdOut_.p, ccdOut_.p + 1, twoPass ? nullptr : ccdHits_.p, counters_.p + 1, twoPass ? 0 : HIT_CAP };
    const RefLists L{ refStart_.p, refItems_.p, refStart_.p + nC1, refItems_.p, refStart_.p + 2 * nC1, refItems_.p };
    auto sweep = [&](int pass) {
        hipLaunchKernelGGL(k_ref_sweep_vertex, dim3(nblk(SWEEP_COOP * (long long)nSVI)), dim3(BLOCK), 0, stream, nSVI, d_SVI.p, d_SF.p, d_SFE.p, x_dev, p_dev, pf,
            d_v2sv.p, refVbox_.p, g, L, alpha, slackness, pass, o, counters_.p);
        // the bound the vertex sweeps left is what the edge pairs' boxes are swept over (SelfCollisionHandler.cpp:1189, 1219)
        if (pass == 0) HIP_CHECK(hipMemcpyAsync(ccdOut_.p + 2, ccdOut_.p, sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream));
        if (nSFE)
            hipLaunchKernelGGL(k_ref_sweep_edge, dim3(nblk(SWEEP_COOP * (long long)nSFE)), dim3(BLOCK), 0, stream, nSFE, d_SFE.p, x_dev, p_dev, pf, d_v2sv.p,
                refVbox_.p, g, L.startE, L.itemsE, alpha, ccdOut_.p + 2, slackness, pass, o, counters_.p);
    };
    sweep(0);
    if (twoPass) sweep(1);
    else hipLaunchKernelGGL(k_ccd_hits_arg, dim3(64), dim3(BLOCK), 0, stream, o);
    unsigned long long h[2];
    int cnt[2];
    // minimum, pair and the two counters in one read-back through mapped memory
    hipLaunchKernelGGL(k_publish_u64x4, dim3(1), dim3(2), 0, stream, (const unsigned long long*)ccdOut_.p, readback_.dev);
    hipLaunchKernelGGL(k_publish_u64, dim3(1), dim3(1), 0, stream, reinterpret_cast<const unsigned long long*>(counters_.p), readback_.dev + 2);
    HIP_CHECK(hipStreamSynchronize(stream));
    h[0] = readback_.p[0];
    h[1] = readback_.p[1];
    std::memcpy(cnt, readback_.p + 2, sizeof(cnt));
    if (!twoPass && cnt[1] > HIT_CAP) { // more hits than the list holds (never seen): the limiting pair by the second run after all
        sweep(1);
        HIP_CHECK(hipMemcpyAsync(h, ccdOut_.p, sizeof(h), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
    }
    if (nCand) *nCand = cnt[0];
    double t;
    std::memcpy(&t, &h[0], sizeof(t));
    if (h[1] == ~0ull || !(t < alpha)) return alpha;
    if (arg3) {
        const int isEE = (int)(h[1] >> 63);
        arg3[0] = isEE ? K_EE : (int)((h[1] >> 31) & 3); // rank 0 / 1 / 2 = K_PP / K_PE / K_PT
        arg3[1] = (int)((h[1] >> 33) & 0x3FFFFFFF);
        arg3[2] = (int)(h[1] & 0x7FFFFFFF);
    }
    return t;
}

bool HipContact::isIntersected(const HipMesh& mesh, const double* x_dev, const int* dbc_dev)
{
    if (!surfaceSet) throw StateError("is_intersected before set_surface");
    const int* pf = pairFlags(mesh.nV, dbc_dev);
    auto pointsInTets = [&]() {
        if (codimPoints.empty() || !mesh.nT) return;
        const dim3 grid(nblk((long long)codimPoints.size() * mesh.nT));
        if (exactPredicates)
            hipLaunchKernelGGL(k_points_in_tets<true>, grid, dim3(BLOCK), 0, stream, (int)codimPoints.size(), d_codimPoints.p, mesh.nT, mesh.d_tet.p, x_dev,
                counters_.p);
        else
            hipLaunchKernelGGL(k_points_in_tets<false>, grid, dim3(BLOCK), 0, stream, (int)codimPoints.size(), d_codimPoints.p, mesh.nT, mesh.d_tet.p, x_dev,
                counters_.p);
    };
    // Round 6: the check runs once or twice per Newton iteration of a contact scene and used to wait for the host three times (bounding box, number of cell
    // entries, result).  Like the constraint-set build it now lays its grid over the box the LAST build or check measured (a stale grid is detected on the
    // device and nothing runs), fills the edge cells into the capacity the last pass needed, and reads flag + stale flag + total + the fresh box back at once.
    for (int attempt = 0; haveBox_ && nSF > 0 && nSFE > 0 && attempt < 3; ++attempt) {
        const int nV = mesh.nV, nb = nblk(nV);
        bboxPartial_.ensure(6 * (size_t)nb + 6);
        double* box_dev = bboxPartial_.p + 6 * (size_t)nb;
        counters_.alloc(16);
        Grid g;
        g.h = mesh.avgEdgeLen;
        long long nCells;
        for (;;) {
            nCells = 1;
            for (int c = 0; c < 3; ++c) {
                g.lo[c] = box_[c] - 2.0 * g.h;
                g.dim[c] = std::max(1, (int)std::floor((box_[3 + c] + 2.0 * g.h - g.lo[c]) / g.h) + 1);
                nCells *= g.dim[c];
            }
            if (nCells <= (1LL << 26)) break;
            g.h *= 1.5;
        }
        if (gridCount_.n < (size_t)nCells + 1) { // (cleared once: every pass counts its cells up and back down)
            gridCount_.ensure((size_t)nCells + 1);
            gridCount_.zeroN(gridCount_.n, stream);
        }
        gridStart_.ensure((size_t)nCells + 1);
        if (gridItems_.n < (size_t)REC * 8 * (size_t)nSFE) gridItems_.ensure((size_t)REC * 8 * (size_t)nSFE);
        const int capItems = (int)std::min<size_t>(gridItems_.n / REC, (size_t)INT_MAX);
        const int* stale = counters_.p + 2;
        readbackInit();
        BuildReadback* rb = reinterpret_cast<BuildReadback*>(readback_.p);
        counters_.zero(stream);
        hipLaunchKernelGGL(k_bbox_partial, dim3(nb), dim3(BLOCK), 0, stream, nV, x_dev, bboxPartial_.p);
        hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(BLOCK), 0, stream, nb, bboxPartial_.p, box_dev, g, 1, counters_.p + 2);
        // (edges only, in cells [0, nCells): the kernel's offset of the edge cells is its nCells argument)
        hipLaunchKernelGGL(k_grid_insert_both, dim3(nblk(nSFE)), dim3(BLOCK), 0, stream, 0, (const int*)nullptr, nSFE, d_SFE.p, x_dev, g, 0, 0.0, 0, capItems, stale,
            gridCount_.p, (const int*)nullptr, (int*)nullptr);
        size_t tmpBytes = 0;
        hipcub::DeviceScan::ExclusiveSum(nullptr, tmpBytes, gridCount_.p, gridStart_.p, (int)nCells + 1, stream);
        if (scanTmp_.n < tmpBytes) scanTmp_.alloc(tmpBytes + tmpBytes / 4);
        hipcub::DeviceScan::ExclusiveSum((void*)scanTmp_.p, tmpBytes, gridCount_.p, gridStart_.p, (int)nCells + 1, stream);
        hipLaunchKernelGGL(k_grid_insert_both, dim3(nblk(nSFE)), dim3(BLOCK), 0, stream, 0, (const int*)nullptr, nSFE, d_SFE.p, x_dev, g, 0, 0.0, 1, capItems, stale,
            gridCount_.p, gridStart_.p, gridItems_.p);
        if (exactPredicates)
            hipLaunchKernelGGL(k_intersect<true>, dim3(nblk(COOP * (long long)nSF)), dim3(BLOCK), 0, stream, nSF, d_SF.p, d_SFE.p, x_dev, pf, g, gridStart_.p,
                gridItems_.p, counters_.p, capItems);
        else
            hipLaunchKernelGGL(k_intersect<false>, dim3(nblk(COOP * (long long)nSF)), dim3(BLOCK), 0, stream, nSF, d_SF.p, d_SFE.p, x_dev, pf, g, gridStart_.p,
                gridItems_.p, counters_.p, capItems);
        pointsInTets();
        hipLaunchKernelGGL(k_publish_narrow, dim3(1), dim3(1), 0, stream, counters_.p, gridStart_.p + nCells, box_dev, reinterpret_cast<BuildReadback*>(readback_.dev));
        HIP_CHECK(hipStreamSynchronize(stream));
        for (int c = 0; c < 6; ++c) box_[c] = rb->box[c];
        if (rb->cnt[2]) continue; // stale grid: nothing ran, once more over the fresh box
        if (rb->cnt[3] > capItems) { // truncated cell lists: grow and repeat
            gridItems_.ensure((size_t)REC * ((size_t)rb->cnt[3] + (size_t)rb->cnt[3] / 4));
            continue;
        }
        return rb->cnt[0] != 0;
    }
    // the general path (first call on a surface, or a box that keeps moving): own bounding box, own cell arrays, three synchronisations
    const GridHost gh = makeGrid(mesh, x_dev, nullptr, 0.0, mesh.avgEdgeLen);
    buildCells(gh, nSFE, 2, d_SFE.p, x_dev, nullptr, 0.0, 0.0, cellCountE_, cellStartE_, cellItemsE_);
    Grid g;
    for (int c = 0; c < 3; ++c) {
        g.lo[c] = gh.lo[c];
        g.dim[c] = gh.dim[c];
    }
    g.h = gh.h;
    counters_.alloc(16);
    counters_.zero(stream);
    if (exactPredicates)
        hipLaunchKernelGGL(k_intersect<true>, dim3(nblk(COOP * (long long)nSF)), dim3(BLOCK), 0, stream, nSF, d_SF.p, d_SFE.p, x_dev, pf, g, cellStartE_.p,
            cellItemsE_.p, counters_.p, INT_MAX);
    else
        hipLaunchKernelGGL(k_intersect<false>, dim3(nblk(COOP * (long long)nSF)), dim3(BLOCK), 0, stream, nSF, d_SF.p, d_SFE.p, x_dev, pf, g, cellStartE_.p,
            cellItemsE_.p, counters_.p, INT_MAX);
    pointsInTets();
    int f[2];
    counters_.download(f, 2, stream);
    return f[0] != 0;
}

// stencils of the current active set closer than dTol, in set order, with their squared distances: evaluated and
// filtered on the device, only the (few) hits come back
void HipContact::closeStencils(const double* x_dev, double dTol, std::vector<std::array<int, 4>>& ids, std::vector<double>& d2, const HipLinSysSolver* lin,
    int* covers)
{
    ids.clear();
    d2.clear();
    if (covers) *covers = 1;
    const int n = nActive_;
    const int nCheck = lin ? nActive_ + nPara_ : 0; // the coverage test of patternCovers rides along (round 6): same sets, same synchronisation
    if (!n && !nCheck) return;
    int cap = std::max<int>(1024, (int)closeIdx_.n);
    readbackInit();
    for (;;) {
        closeIdx_.ensure((size_t)cap);
        closeVal_.ensure((size_t)cap);
        counters_.alloc(16);
        counters_.zero(stream);
        if (n) hipLaunchKernelGGL(k_close_stencils, dim3(nblk(n)), dim3(BLOCK), 0, stream, n, d_active.p, x_dev, dTol, cap, closeIdx_.p, closeVal_.p, counters_.p);
        if (nCheck) {
            ContactView cv{ nActive_, nPara_, d_active.p, d_para.p, d_paraEIEJ.p, d_SFE.p, nullptr, d_xRest.p };
            CsrView m{ lin->d_ia.p, lin->d_ja.p };
            hipLaunchKernelGGL(k_pattern_check, dim3(nblk(nCheck)), dim3(BLOCK), 0, stream, cv, m, counters_.p + 2);
        }
        hipLaunchKernelGGL(k_publish_u64x4, dim3(1), dim3(2), 0, stream, reinterpret_cast<const unsigned long long*>(counters_.p), readback_.dev); // counters 0 .. 3
        HIP_CHECK(hipStreamSynchronize(stream));
        const int* h = reinterpret_cast<const int*>(readback_.p);
        const int cnt = h[0];
        if (covers) *covers = h[2] ? 0 : 1;
        if (cnt > cap) {
            cap = cnt + cnt / 4;
            continue;
        }
        if (!cnt) return;
        std::vector<int> idx(cnt);
        std::vector<double> val(cnt);
        closeIdx_.download(idx.data(), cnt, stream);
        closeVal_.download(val.data(), cnt, stream);
        HIP_CHECK(hipStreamSynchronize(stream));
        std::vector<int> order(cnt);
        for (int i = 0; i < cnt; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return idx[a] < idx[b]; }); // set order (the kernel appends atomically)
        // the tuples of the hits: gather from the device set
        syncHost();
        for (int k : order) {
            ids.push_back(active[idx[k]]);
            d2.push_back(val[k]);
        }
        return;
    }
}

void HipContact::evalStencils(const std::vector<std::array<int, 4>>& ids, const double* x_dev, std::vector<double>& d2)
{
    const int n = (int)ids.size();
    d2.assign(n, 0.0);
    if (!n) return;
    std::vector<int> flat(4 * (size_t)n);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 4; ++k) flat[4 * (size_t)i + k] = ids[i][k];
    d_ids_.uploadGrow(flat, stream);
    d_vals_.ensure(n);
    hipLaunchKernelGGL(k_eval_stencils, dim3(nblk(n)), dim3(BLOCK), 0, stream, n, d_ids_.p, x_dev, d_vals_.p);
    d_vals_.download(d2.data(), n, stream);
}

} // namespace ipcgpu
