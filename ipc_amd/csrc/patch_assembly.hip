// Patch-parallel, atomic-free Newton assembly (see patch_assembly.h).
#include "patch_assembly.h"
#include "hip_ipc.h"
#include "nh_device.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>

namespace ipcgpu {

namespace {

constexpr int BLOCK_MAX = 512; // workgroup sizes: 256 (patches of <= 256 elements, two workgroups per CU) or 512 (<= 512 elements, one per CU: less halo)
constexpr int NF = 38; // staged doubles per element (one 304-byte record: 76 dwords = 12 mod 64, so records spread over the LDS banks):
                        // U 9, Ad 6, Bd 6, Bo 3, beta_0..3 12, projection mask (int), pad
constexpr int CHUNK = 4; // contributions per phase-2 lane
constexpr int MAXSEG = 8; // chunks per block (segmented reduction over 1, 2, 4 lanes of a 16-lane row)
using namespace dev;

// lane i <- lane i + N inside its 16-lane row (v_mov_b32 row_shl:N), zero shifted in
template <int N>
__device__ __forceinline__ double row_shl(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x100 + N, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x100 + N, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

#ifdef IPCGPU_ASM_WAVES3 // experiment build (tools/): three waves per SIMD at the price of spills
#define ASM_OCC __attribute__((amdgpu_waves_per_eu(3, 3)))
#else
#define ASM_OCC
#endif
template <bool HESS, int BLOCK>
__global__ __launch_bounds__(BLOCK) ASM_OCC void k_assemble_patch(ElemView v, PatchView pv, int patchBegin, int tcap, int ncap, double coef,
    int projectDBC, double* __restrict__ grad, double* __restrict__ a, int probe, const int* __restrict__ patchList)
{
    extern __shared__ __align__(16) double lds[];
    double* stage = lds; // [tcap][NF]: phase 2 reads a record with 128-bit LDS loads, the two beta rows by dynamic offset
    double* gacc = lds + (size_t)NF * tcap; // [3 * ncap]
    // record slot 36 (as int): projection mask (bit k: node k projected, bit 4: active)
    const int p = patchList ? patchList[blockIdx.x] : patchBegin + blockIdx.x; // a list: the patches this rank owns rows in (owner-computes sharding)
    const int n0 = pv.nodePtr[p], nOwned = pv.nodePtr[p + 1] - n0;
    const int t0 = pv.tetPtr[p], nTets = pv.tetPtr[p + 1] - t0;
    const int tid = threadIdx.x;
    if (grad) {
        for (int i = tid; i < 3 * nOwned; i += BLOCK) gacc[i] = 0.0;
        __syncthreads();
    }
    // ---- phase 1: generators of every element touching the patch
    for (int base = 0; base < nTets; base += BLOCK) { // (uniform trip count: the loop body holds barriers)
        const int tl = base + tid;
        const bool act = tl < nTets;
        uint16_t gs[4] = { 0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu };
        double fb[12]; // forces of this lane's element on its four nodes
#pragma unroll
        for (int k = 0; k < 12; ++k) fb[k] = 0.0;
        if (act) {
        const int inst = t0 + tl;
#pragma unroll
        for (int k = 0; k < 4; ++k) gs[k] = grad ? pv.gradSlot[(size_t)k * pv.totalTets + inst] : (uint16_t)0xFFFF;
        ElemGen g;
        element_generators(v, pv.tets[inst], coef, projectDBC, grad != nullptr, HESS, [&](int k, int i, double val) { fb[3 * k + i] = val; }, g);
        if (HESS) {
            int mask = g.active ? 16 : 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (projected_dbc(g.dtype[k], projectDBC)) mask |= (1 << k);
            double* e = stage + (size_t)tl * NF;
            *reinterpret_cast<int*>(e + 36) = mask;
            if (g.active) {
#pragma unroll
                for (int i = 0; i < 9; ++i) e[i] = g.U[i];
#pragma unroll
                for (int i = 0; i < 6; ++i) e[9 + i] = g.Ad[i];
#pragma unroll
                for (int i = 0; i < 6; ++i) e[15 + i] = g.Bd[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) e[21 + i] = g.Bo[i];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int q = 0; q < 3; ++q) e[24 + 3 * k + q] = g.beta[k][q];
            }
        }
        }
        if (grad) {
            // nodal forces into the LDS accumulators, ONE WAVE AT A TIME and in instruction order inside a wave: the order of the
            // additions to a node is then a function of the plan alone, and the gradient comes out with the same bits on every run
            // (ds_add_f64 across waves retires in scheduling order -- measured: last-bit noise run to run)
            for (int w = 0; w < BLOCK / 64; ++w) {
                if ((tid >> 6) == w && act) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (gs[k] != 0xFFFFu) {
#pragma unroll
                            for (int i = 0; i < 3; ++i) atomicAdd(&gacc[3 * (int)gs[k] + i], fb[3 * k + i]); // ds_add_f64, 12 per element
                        }
                }
                __syncthreads();
            }
        }
    }
    __syncthreads();
    // ---- phase 2: one lane per (destination block, <= CHUNK contributions)
    if (HESS && a && probe != 1) {
        // Every global read of a pass is issued before the pass's arithmetic (and the next pass's reads before this
        // pass's): a dependent item -> range -> contribution -> LDS chain costs four memory latencies per pass, which two
        // waves per SIMD cannot hide.
        const int i0 = pv.itemPtr[p], i1 = pv.itemPtr[p + 1];
        const int4 hdrPad = make_int4(-1, 1 << 16, 0, 0); // padding: segLen 1, segPos 0, no contributions
        const uint4 c4Pad = make_uint4(0, 0, 0, 0);
        int4 hdr = hdrPad;
        uint4 c4 = c4Pad;
        if (i0 + tid < i1) {
            hdr = pv.itemHdr[i0 + tid];
            c4 = pv.itemC4[i0 + tid];
        }
        for (int base = i0; base < i1; base += BLOCK) {
            int4 hdrN = hdrPad;
            uint4 c4N = c4Pad;
            if (base + BLOCK + tid < i1) {
                hdrN = pv.itemHdr[base + BLOCK + tid];
                c4N = pv.itemC4[base + BLOCK + tid];
            }
            const int p0 = hdr.x, rowNode = hdr.z;
            const uint32_t meta = (uint32_t)hdr.y;
            const int nC = (int)(meta >> 25);
            int rowDbc = 0;
            double rowMass = 0.0;
            if (p0 >= 0 && ((meta >> 20) & 15) == 0) {
                rowDbc = v.dbc[rowNode];
                rowMass = v.mass[rowNode];
            }
            double S[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < 3; ++r) S[i][r] = 0.0;
            auto accumulate = [&](uint32_t cw) {
                const int tl = cw & 0xFFFF, ka = (cw >> 16) & 3, kc = (cw >> 18) & 3;
                const double* e = stage + (size_t)tl * NF;
                const int mask = *reinterpret_cast<const int*>(e + 36);
                if (!(mask & 16) || (mask & ((1 << ka) | (1 << kc)))) return; // IglUtils.hpp:45-53: projected rows / columns dropped
                const double2* e2 = reinterpret_cast<const double2*>(e);
                double rec[24];
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    const double2 t = e2[i];
                    rec[2 * i] = t.x;
                    rec[2 * i + 1] = t.y;
                }
                const double *U = rec, *Ad = rec + 9, *Bd = rec + 15, *Bo = rec + 21;
                const double *pa = e + 24 + 3 * ka, *pc = e + 24 + 3 * kc;
                const double ba[3] = { pa[0], pa[1], pa[2] }, bc[3] = { pc[0], pc[1], pc[2] };
                double H[3][3];
                pair_block(U, ba, bc, Ad, Bd, Bo, H);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int r = 0; r < 3; ++r) S[i][r] += H[i][r];
            };
            if (nC > 0) accumulate(c4.x);
            if (nC > 1) accumulate(c4.y);
            if (nC > 2) accumulate(c4.z);
            if (nC > 3) accumulate(c4.w);
            for (int c = 4; c < nC; ++c) accumulate(pv.contrib[hdr.w + c]); // blocks with more than 4 * MAXSEG contributions
            // segmented reduction over the (<= MAXSEG) chunks of a block, which sit in adjacent lanes of one 16-lane row:
            // DPP row shifts (lanes shifted in from outside the row read 0 and are masked out anyway)
            const int segLen = (meta >> 16) & 15, segPos = (meta >> 20) & 15;
            const bool m1 = segPos + 1 < segLen, m2 = segPos + 2 < segLen, m4 = segPos + 4 < segLen;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const double t1 = row_shl<1>(S[i][r]);
                    S[i][r] += m1 ? t1 : 0.0;
                    const double t2 = row_shl<2>(S[i][r]);
                    S[i][r] += m2 ? t2 : 0.0;
                    const double t4 = row_shl<4>(S[i][r]);
                    S[i][r] += m4 ? t4 : 0.0;
                }
            if (p0 >= 0 && segPos == 0) {
                const int L = meta & 0xFFFF;
                const bool isDiag = (meta >> 24) & 1;
                const bool proj = projected_dbc(rowDbc, projectDBC);
                if (isDiag) {
                    const double m = rowMass;
                    if (proj) { // Optimizer.cpp:3654-3663
                        S[0][0] = S[1][1] = S[2][2] = 1.0;
                        S[0][1] = S[0][2] = S[1][2] = 0.0;
                    }
                    else { // :3641-3649
                        S[0][0] += m;
                        S[1][1] += m;
                        S[2][2] += m;
                    }
                    a[p0 + 0] = S[0][0];
                    a[p0 + 1] = S[0][1];
                    a[p0 + 2] = S[0][2];
                    a[p0 + L + 0] = S[1][1];
                    a[p0 + L + 1] = S[1][2];
                    a[p0 + 2 * L - 1] = S[2][2];
                }
                else {
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const int rowOff = (r == 0) ? 0 : (r == 1 ? (L - 1) : (2 * L - 3));
#pragma unroll
                        for (int c = 0; c < 3; ++c) a[p0 + rowOff + c] = proj ? 0.0 : S[r][c];
                    }
                }
            }
            hdr = hdrN;
            c4 = c4N;
        }
    }
    // ---- gradient flush: elastic forces + m (x - xTilde)  (Optimizer.cpp:3438-3450)
    if (grad) {
        for (int i = tid; i < 3 * nOwned; i += BLOCK) {
            const int ln = i / 3, c = i - 3 * ln;
            const int node = pv.nodes[n0 + ln];
            double g = gacc[i];
            if (!projected_dbc(v.dbc[node], projectDBC)) g += v.mass[node] * (v.x[3 * (size_t)node + c] - v.xTilde[3 * (size_t)node + c]);
            grad[3 * (size_t)node + c] = g;
        }
    }
}

inline uint32_t morton3(uint32_t x, uint32_t y, uint32_t z)
{
    auto spread = [](uint32_t v) {
        v &= 0x3FF;
        v = (v | (v << 16)) & 0x030000FF;
        v = (v | (v << 8)) & 0x0300F00F;
        v = (v | (v << 4)) & 0x030C30C3;
        v = (v | (v << 2)) & 0x09249249;
        return v;
    };
    return spread(x) | (spread(y) << 1) | (spread(z) << 2);
}

} // namespace

void PatchPlan::build(const HipMesh& mesh, const HipLinSysSolver& lin, hipStream_t s)
{
    valid = false;
    const int nV = mesh.nV, nT = mesh.nT;
    if (lin.rowBase.empty() || nT == 0) return;
    // elements per patch: 256 fills a 256-thread workgroup and has the least halo; a mesh that is only a round or two of resident workgroups (two per CU)
    // does better with 224 (mat150: 45.9 us against 47.8 -- more, shorter workgroups in the second round; mat433: 0.361 ms against 0.307,
    // profiles/r04_assembly_patch_size_ab.txt)
    const int tetCap = (nT < 300000) ? 224 : 256;
    std::vector<int> owner, localIdx; // filled with the patches (fresh topology only): who owns a node, and where in its patch
    const bool freshTopo = !(topo.mesh == (const void*)&mesh && topo.version == mesh.featuresVersion && topo.nV == nV && topo.nT == nT && topo.tetCap == tetCap);
    if (freshTopo) {
        topo.mesh = &mesh;
        topo.nV = nV;
        topo.nT = nT;
        topo.tetCap = tetCap;
        topo.version = mesh.featuresVersion;
        // node -> incident elements (with the local index of the node inside the element)
        std::vector<int>&vtPtr = topo.vtPtr, &vt = topo.vt, &vtLoc = topo.vtLoc;
        vtPtr.assign(nV + 1, 0);
        vt.assign(4 * (size_t)nT, 0);
        vtLoc.assign(4 * (size_t)nT, 0);
        for (int t = 0; t < nT; ++t)
            for (int k = 0; k < 4; ++k) vtPtr[mesh.F[t + (size_t)nT * k] + 1]++;
        for (int v = 0; v < nV; ++v) vtPtr[v + 1] += vtPtr[v];
        {
            std::vector<int> pos(vtPtr.begin(), vtPtr.end() - 1);
            for (int t = 0; t < nT; ++t)
                for (int k = 0; k < 4; ++k) {
                    const int v = mesh.F[t + (size_t)nT * k];
                    vt[pos[v]] = t;
                    vtLoc[pos[v]] = k;
                    pos[v]++;
                }
        }
        // Morton order of the rest positions (cubic cells so that thin directions collapse)
        double ext = 0;
        for (int c = 0; c < 3; ++c) ext = std::max(ext, mesh.bboxHi[c] - mesh.bboxLo[c]);
        if (!(ext > 0)) ext = 1;
        std::vector<std::pair<uint32_t, int>> keyed(nV);
        for (int v = 0; v < nV; ++v) {
            uint32_t q[3];
            for (int c = 0; c < 3; ++c) q[c] = (uint32_t)std::min(1023.0, std::max(0.0, (mesh.V_rest[v + (size_t)nV * c] - mesh.bboxLo[c]) / ext * 1023.0));
            keyed[v] = { morton3(q[0], q[1], q[2]), v };
        }
        std::sort(keyed.begin(), keyed.end());
        // greedy patches: consecutive Morton nodes while the touched elements fit one phase-1 round
        std::vector<int>&hNodePtr = topo.nodePtr, &hNodes = topo.nodes, &hTetPtr = topo.tetPtr, &hTets = topo.tets;
        hNodePtr.assign(1, 0);
        hTetPtr.assign(1, 0);
        hNodes.clear();
        hTets.clear();
        std::vector<int> mark(nT, -1);
        std::vector<int> curTets;
        owner.assign(nV, -1);
        localIdx.assign(nV, 0);
        int pid = 0;
        auto closePatch = [&]() {
            std::sort(curTets.begin(), curTets.end());
            hTets.insert(hTets.end(), curTets.begin(), curTets.end());
            hTetPtr.push_back((int)hTets.size());
            hNodePtr.push_back((int)hNodes.size());
            curTets.clear();
            ++pid;
        };
        for (int i = 0; i < nV; ++i) {
            const int v = keyed[i].second;
            int fresh = 0;
            for (int k = vtPtr[v]; k < vtPtr[v + 1]; ++k)
                if (mark[vt[k]] != pid) ++fresh;
            if ((int)hNodes.size() > hNodePtr.back() && (int)curTets.size() + fresh > tetCap) closePatch();
            for (int k = vtPtr[v]; k < vtPtr[v + 1]; ++k)
                if (mark[vt[k]] != pid) {
                    mark[vt[k]] = pid;
                    curTets.push_back(vt[k]);
                }
            if ((int)curTets.size() > 65535) throw StateError("a single node touches more elements than the 16-bit element slot can hold");
            owner[v] = pid;
            localIdx[v] = (int)hNodes.size() - hNodePtr.back();
            hNodes.push_back(v);
        }
        if ((int)hNodes.size() > hNodePtr.back()) closePatch();
    }
    const std::vector<int>&vtPtr = topo.vtPtr, &vt = topo.vt, &vtLoc = topo.vtLoc;
    const std::vector<int>&hNodePtr = topo.nodePtr, &hNodes = topo.nodes, &hTetPtr = topo.tetPtr, &hTets = topo.tets;
    nPatches = (int)hNodePtr.size() - 1;
    totalTets = (long long)hTets.size();
    haloFactor = (double)totalTets / nT;
    maxTets = maxNodes = 0;
    for (int p = 0; p < nPatches; ++p) {
        maxTets = std::max(maxTets, hTetPtr[p + 1] - hTetPtr[p]);
        maxNodes = std::max(maxNodes, hNodePtr[p + 1] - hNodePtr[p]);
    }
    // gradient slots + phase-2 work items: per patch, independent of each other -- a few threads, each with its own output
    std::vector<uint16_t> hGrad;
    if (freshTopo) hGrad.assign(4 * (size_t)totalTets, 0xFFFFu);
    struct Part {
        std::vector<int> p0, row, cptr, itemEnd; // itemEnd[patch - first]: items of this part up to and including the patch
        std::vector<uint32_t> meta, contrib;
    };
    const int nThreads = std::max(1, std::min(8, std::min((int)std::thread::hardware_concurrency(), nPatches / 64 + 1)));
    std::vector<Part> parts(nThreads);
    auto work = [&](int ti) {
        Part& P = parts[ti];
        const int pBeg = (int)((long long)nPatches * ti / nThreads), pEnd = (int)((long long)nPatches * (ti + 1) / nThreads);
        std::vector<int> tetLocal(nT, -1);
        std::vector<uint32_t> cl;
        auto pushItem = [&](int p0, int L, int segLen, int segPos, bool isDiag, int row, const uint32_t* cb, const uint32_t* ce) {
            P.p0.push_back(p0);
            P.meta.push_back((uint32_t)L | ((uint32_t)segLen << 16) | ((uint32_t)segPos << 20) | ((uint32_t)(isDiag ? 1 : 0) << 24));
            P.row.push_back(row);
            P.cptr.push_back((int)P.contrib.size());
            P.contrib.insert(P.contrib.end(), cb, ce);
        };
        for (int p = pBeg; p < pEnd; ++p) {
            for (int inst = hTetPtr[p]; inst < hTetPtr[p + 1]; ++inst) {
                const int t = hTets[inst];
                tetLocal[t] = inst - hTetPtr[p];
                if (freshTopo)
                    for (int k = 0; k < 4; ++k) {
                        const int vv = mesh.F[t + (size_t)nT * k];
                        if (owner[vv] == p) hGrad[(size_t)k * totalTets + inst] = (uint16_t)localIdx[vv];
                    }
            }
            const int itemStart = (int)P.p0.size();
            auto addBlock = [&](int p0, int L, bool isDiag, int row) {
                const int n = (int)cl.size();
                const int chunk = std::max(CHUNK, (n + MAXSEG - 1) / MAXSEG);
                const int segLen = std::max(1, (n + chunk - 1) / chunk);
                // a block's chunks must stay inside one 16-lane DPP row
                const int posInRow = ((int)P.p0.size() - itemStart) % 16;
                if (posInRow + segLen > 16)
                    for (int pad = posInRow; pad < 16; ++pad) pushItem(-1, 0, 1, 0, false, 0, nullptr, nullptr);
                for (int sgi = 0; sgi < segLen; ++sgi) {
                    const int b = sgi * chunk, e = std::min(n, b + chunk);
                    pushItem(p0, L, segLen, sgi, isDiag, row, cl.data() + b, cl.data() + e);
                }
            };
            for (int j = hNodePtr[p]; j < hNodePtr[p + 1]; ++j) {
                const int vv = hNodes[j];
                const int L = lin.rowLen[vv], base = lin.rowBase[vv];
                // diagonal block: every incident element
                cl.clear();
                for (int k = vtPtr[vv]; k < vtPtr[vv + 1]; ++k)
                    cl.push_back((uint32_t)tetLocal[vt[k]] | ((uint32_t)vtLoc[k] << 16) | ((uint32_t)vtLoc[k] << 18));
                addBlock(base, L, true, vv);
                // off-diagonal blocks (vv, n), n > vv ascending: elements containing both
                const int nUp = (L - 3) / 3;
                for (int r = 0; r < nUp; ++r) {
                    const int n = lin.ja[base + 3 + 3 * r] / 3;
                    cl.clear();
                    for (int k = vtPtr[vv]; k < vtPtr[vv + 1]; ++k) {
                        const int t = vt[k];
                        for (int kk = 0; kk < 4; ++kk)
                            if (mesh.F[t + (size_t)nT * kk] == n)
                                cl.push_back((uint32_t)tetLocal[t] | ((uint32_t)vtLoc[k] << 16) | ((uint32_t)kk << 18));
                    }
                    addBlock(base + 3 + 3 * r, L, false, vv);
                }
            }
            P.itemEnd.push_back((int)P.p0.size());
            for (int inst = hTetPtr[p]; inst < hTetPtr[p + 1]; ++inst) tetLocal[hTets[inst]] = -1;
        }
    };
    if (nThreads == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int ti = 0; ti < nThreads; ++ti) pool.emplace_back(work, ti);
        for (auto& th : pool) th.join();
    }
    // concatenate the parts (item and contribution offsets shift by what came before)
    size_t nItems = 0, nContrib = 0;
    for (const Part& P : parts) {
        nItems += P.p0.size();
        nContrib += P.contrib.size();
    }
    totalItems = (long long)nItems;
    totalContribs = (long long)nContrib;
    std::vector<int> hItemPtr{ 0 };
    hItemPtr.reserve(nPatches + 1);
    if (hHdrPin.n < nItems) {
        hHdrPin.alloc(nItems + nItems / 2 + 16);
        hC4Pin.alloc(nItems + nItems / 2 + 16);
    }
    if (hContribPin.n < nContrib + 1) hContribPin.alloc(nContrib + nContrib / 2 + 16);
    int4* hHdr = hHdrPin.p;
    uint4* hC4 = hC4Pin.p;
    uint32_t* hContrib = hContribPin.p;
    size_t itemBase = 0, cBase = 0;
    for (const Part& P : parts) {
        for (size_t i = 0; i < P.p0.size(); ++i) {
            const int c0 = P.cptr[i], n = (int)((i + 1 < P.p0.size() ? (size_t)P.cptr[i + 1] : P.contrib.size()) - (size_t)c0);
            if (n > 127) throw StateError("more than 127 element contributions in one chunk of a CSR block");
            hHdr[itemBase + i] = make_int4(P.p0[i], (int)(P.meta[i] | ((uint32_t)n << 25)), P.row[i], (int)(cBase + c0));
            uint32_t w[4] = { 0, 0, 0, 0 };
            for (int k = 0; k < 4 && k < n; ++k) w[k] = P.contrib[c0 + k];
            hC4[itemBase + i] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        for (int e : P.itemEnd) hItemPtr.push_back((int)(itemBase + e));
        if (!P.contrib.empty()) std::memcpy(hContrib + cBase, P.contrib.data(), P.contrib.size() * sizeof(uint32_t));
        cBase += P.contrib.size();
        itemBase += P.p0.size();
    }
    if (nContrib == 0) hContrib[0] = 0;
    if (freshTopo) {
        nodePtr.upload(hNodePtr, s);
        nodes.upload(hNodes, s);
        tetPtr.upload(hTetPtr, s);
        tets.upload(hTets, s);
        gradSlot.upload(hGrad, s);
    }
    itemPtr.uploadGrow(hItemPtr, s);
    itemHdr.ensure(nItems);
    itemC4.ensure(nItems);
    contrib.ensure(std::max<size_t>(nContrib, 1));
    if (nItems) {
        HIP_CHECK(hipMemcpyAsync(itemHdr.p, hHdr, nItems * sizeof(int4), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(itemC4.p, hC4, nItems * sizeof(uint4), hipMemcpyHostToDevice, s));
    }
    HIP_CHECK(hipMemcpyAsync(contrib.p, hContrib, std::max<size_t>(nContrib, 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));
    valid = true;
}

PatchView PatchPlan::view() const
{
    PatchView pv;
    pv.nPatches = nPatches;
    pv.nodePtr = nodePtr.p;
    pv.nodes = nodes.p;
    pv.tetPtr = tetPtr.p;
    pv.tets = tets.p;
    pv.gradSlot = gradSlot.p;
    pv.totalTets = totalTets;
    pv.itemPtr = itemPtr.p;
    pv.itemHdr = itemHdr.p;
    pv.itemC4 = itemC4.p;
    pv.contrib = contrib.p;
    return pv;
}

size_t PatchPlan::ldsBytes() const
{
    const size_t tcap = (size_t)((maxTets + 63) / 64 * 64);
    return sizeof(double) * (NF * tcap + 3 * (size_t)maxNodes);
}

void launch_assemble_patches(const ElemView& v, const PatchPlan& plan, int patchBegin, int patchEnd, double coef, int projectDBC,
    double* grad, double* a, hipStream_t s, const int* patchList)
{
    const int n = patchEnd - patchBegin; // with a list: its length (patchBegin = 0)
    if (n <= 0) return;
    const size_t lds = plan.ldsBytes();
    const int tcap = (plan.maxTets + 63) / 64 * 64;
    const bool wide = plan.maxTets > 256; // patches of up to 512 elements: 512-thread workgroups
    static size_t attrSet[2] = { 0, 0 };
    if (lds > 64 * 1024 && lds > attrSet[wide]) {
        if (wide) {
            HIP_CHECK(hipFuncSetAttribute((const void*)k_assemble_patch<true, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIP_CHECK(hipFuncSetAttribute((const void*)k_assemble_patch<false, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        else {
            HIP_CHECK(hipFuncSetAttribute((const void*)k_assemble_patch<true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIP_CHECK(hipFuncSetAttribute((const void*)k_assemble_patch<false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        attrSet[wide] = lds;
    }
    const PatchView pv = plan.view();
    const int probe = 0;
    if (wide) {
        if (a)
            hipLaunchKernelGGL((k_assemble_patch<true, 512>), dim3(n), dim3(512), lds, s, v, pv, patchBegin, tcap, plan.maxNodes, coef, projectDBC, grad, a, probe, patchList);
        else
            hipLaunchKernelGGL((k_assemble_patch<false, 512>), dim3(n), dim3(512), lds, s, v, pv, patchBegin, tcap, plan.maxNodes, coef, projectDBC, grad, a, probe, patchList);
    }
    else {
        if (a)
            hipLaunchKernelGGL((k_assemble_patch<true, 256>), dim3(n), dim3(256), lds, s, v, pv, patchBegin, tcap, plan.maxNodes, coef, projectDBC, grad, a, probe, patchList);
        else
            hipLaunchKernelGGL((k_assemble_patch<false, 256>), dim3(n), dim3(256), lds, s, v, pv, patchBegin, tcap, plan.maxNodes, coef, projectDBC, grad, a, probe, patchList);
    }
}

} // namespace ipcgpu
