// Patch-parallel, atomic-free Newton assembly (see patch_assembly.h).
#include "patch_assembly.h"
#include "hip_ipc.h"
#include "nh_device.h"
#include <algorithm>
#include <cstdlib>
#include <numeric>

namespace ipcgpu {

namespace {

constexpr int BLOCK = 256;
constexpr int NF = 33; // staged doubles per element: U 9, beta_1..3 9, Ad 6, Bd 6, Bo 3
constexpr int CHUNK = 4; // contributions per phase-2 lane
constexpr int MAXSEG = 8; // chunks per block (wave-level segmented reduction over 1, 2, 4 lanes)
using namespace dev;

template <bool HESS>
__global__ __launch_bounds__(BLOCK) void k_assemble_patch(ElemView v, PatchView pv, int patchBegin, int tcap, int ncap, double coef,
    int projectDBC, double* __restrict__ grad, double* __restrict__ a, int probe)
{
    extern __shared__ double lds[];
    double* stage = lds; // [NF][tcap]
    double* gacc = lds + (size_t)NF * tcap; // [3 * ncap]
    int* pm = reinterpret_cast<int*>(gacc + 3 * ncap); // [tcap] projection mask (bit k: node k projected, bit 4: active)
    const int p = patchBegin + blockIdx.x;
    const int n0 = pv.nodePtr[p], nOwned = pv.nodePtr[p + 1] - n0;
    const int t0 = pv.tetPtr[p], nTets = pv.tetPtr[p + 1] - t0;
    const int tid = threadIdx.x;
    if (grad) {
        for (int i = tid; i < 3 * nOwned; i += BLOCK) gacc[i] = 0.0;
        __syncthreads();
    }
    // ---- phase 1: generators of every element touching the patch
    for (int tl = tid; tl < nTets; tl += BLOCK) {
        const int inst = t0 + tl;
        uint16_t gs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) gs[k] = grad ? pv.gradSlot[(size_t)k * pv.totalTets + inst] : (uint16_t)0xFFFF;
        ElemGen g;
        element_generators(v, pv.tets[inst], coef, projectDBC, grad != nullptr, HESS,
            [&](int k, int i, double val) {
                if (gs[k] != 0xFFFFu) atomicAdd(&gacc[3 * (int)gs[k] + i], val); // ds_add_f64, 12 per element
            },
            g);
        if (HESS) {
            int mask = g.active ? 16 : 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (projected_dbc(g.dtype[k], projectDBC)) mask |= (1 << k);
            pm[tl] = mask;
            if (g.active) {
#pragma unroll
                for (int i = 0; i < 9; ++i) stage[(size_t)i * tcap + tl] = g.U[i];
#pragma unroll
                for (int k = 1; k < 4; ++k)
#pragma unroll
                    for (int q = 0; q < 3; ++q) stage[(size_t)(9 + 3 * (k - 1) + q) * tcap + tl] = g.beta[k][q];
#pragma unroll
                for (int i = 0; i < 6; ++i) stage[(size_t)(18 + i) * tcap + tl] = g.Ad[i];
#pragma unroll
                for (int i = 0; i < 6; ++i) stage[(size_t)(24 + i) * tcap + tl] = g.Bd[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) stage[(size_t)(30 + i) * tcap + tl] = g.Bo[i];
            }
        }
    }
    __syncthreads();
    // ---- phase 2: one lane per (destination block, <= CHUNK contributions)
    if (HESS && a && probe != 1) {
        const int i0 = pv.itemPtr[p], i1 = pv.itemPtr[p + 1];
        for (int base = i0; base < i1; base += BLOCK) {
            const int it = base + tid;
            const int p0 = (it < i1) ? pv.itemP0[it] : -1;
            double S[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < 3; ++r) S[i][r] = 0.0;
            uint32_t meta = 1u << 16; // segLen 1, segPos 0
            int rowNode = 0;
            if (p0 >= 0) {
                meta = pv.itemMeta[it];
                rowNode = pv.itemRow[it];
                const int c0 = pv.itemCPtr[it], c1 = pv.itemCPtr[it + 1];
                for (int c = c0; c < c1; ++c) {
                    const uint32_t cw = pv.contrib[c];
                    const int tl = cw & 0xFFFF, ka = (cw >> 16) & 3, kc = (cw >> 18) & 3;
                    const int mask = pm[tl];
                    if (!(mask & 16) || (mask & ((1 << ka) | (1 << kc)))) continue; // IglUtils.hpp:45-53: projected rows / columns dropped
                    double U[9], ba[3], bc[3], Ad[6], Bd[6], Bo[3];
#pragma unroll
                    for (int i = 0; i < 9; ++i) U[i] = stage[(size_t)i * tcap + tl];
                    double b1[3], b2[3], b3[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        b1[q] = stage[(size_t)(9 + q) * tcap + tl];
                        b2[q] = stage[(size_t)(12 + q) * tcap + tl];
                        b3[q] = stage[(size_t)(15 + q) * tcap + tl];
                    }
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const double b0 = -b1[q] - b2[q] - b3[q];
                        ba[q] = ka == 0 ? b0 : (ka == 1 ? b1[q] : (ka == 2 ? b2[q] : b3[q]));
                        bc[q] = kc == 0 ? b0 : (kc == 1 ? b1[q] : (kc == 2 ? b2[q] : b3[q]));
                    }
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        Ad[i] = stage[(size_t)(18 + i) * tcap + tl];
                        Bd[i] = stage[(size_t)(24 + i) * tcap + tl];
                    }
#pragma unroll
                    for (int i = 0; i < 3; ++i) Bo[i] = stage[(size_t)(30 + i) * tcap + tl];
                    double H[3][3];
                    pair_block(U, ba, bc, Ad, Bd, Bo, H);
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int r = 0; r < 3; ++r) S[i][r] += H[i][r];
                }
            }
            // wave-level segmented reduction over the (<= 4) chunks of a block, which sit in adjacent lanes
            const int segLen = (meta >> 16) & 15, segPos = (meta >> 20) & 15;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    double t1 = __shfl_down(S[i][r], 1, 64);
                    if (segPos + 1 < segLen) S[i][r] += t1;
                    double t2 = __shfl_down(S[i][r], 2, 64);
                    if (segPos + 2 < segLen) S[i][r] += t2;
                    double t4 = __shfl_down(S[i][r], 4, 64);
                    if (segPos + 4 < segLen) S[i][r] += t4;
                }
            if (p0 >= 0 && segPos == 0) {
                const int L = meta & 0xFFFF;
                const bool isDiag = (meta >> 24) & 1;
                const bool proj = projected_dbc(v.dbc[rowNode], projectDBC);
                if (isDiag) {
                    const double m = v.mass[rowNode];
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
    int tetCap = BLOCK;
    if (const char* e = std::getenv("IPCGPU_PATCH_TETS")) tetCap = std::max(32, std::min(BLOCK, std::atoi(e)));
    // node -> incident elements (with the local index of the node inside the element)
    std::vector<int> vtPtr(nV + 1, 0), vt(4 * (size_t)nT), vtLoc(4 * (size_t)nT);
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
    std::vector<int> hNodePtr{ 0 }, hNodes, hTetPtr{ 0 }, hTets;
    std::vector<int> mark(nT, -1), owner(nV, -1), localIdx(nV, 0);
    std::vector<int> curTets;
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
    nPatches = pid;
    totalTets = (long long)hTets.size();
    haloFactor = (double)totalTets / nT;
    maxTets = maxNodes = 0;
    for (int p = 0; p < nPatches; ++p) {
        maxTets = std::max(maxTets, hTetPtr[p + 1] - hTetPtr[p]);
        maxNodes = std::max(maxNodes, hNodePtr[p + 1] - hNodePtr[p]);
    }
    // gradient slots + phase-2 work items
    std::vector<uint16_t> hGrad(4 * (size_t)totalTets, 0xFFFFu);
    std::vector<int> hItemPtr{ 0 }, hP0, hRow, hCPtr;
    std::vector<uint32_t> hMeta, hContrib;
    std::vector<int> tetLocal(nT, -1);
    auto pushItem = [&](int p0, int L, int segLen, int segPos, bool isDiag, int row, const uint32_t* cb, const uint32_t* ce) {
        hP0.push_back(p0);
        hMeta.push_back((uint32_t)L | ((uint32_t)segLen << 16) | ((uint32_t)segPos << 20) | ((uint32_t)(isDiag ? 1 : 0) << 24));
        hRow.push_back(row);
        hCPtr.push_back((int)hContrib.size());
        hContrib.insert(hContrib.end(), cb, ce);
    };
    std::vector<uint32_t> cl;
    for (int p = 0; p < nPatches; ++p) {
        for (int inst = hTetPtr[p]; inst < hTetPtr[p + 1]; ++inst) {
            const int t = hTets[inst];
            tetLocal[t] = inst - hTetPtr[p];
            for (int k = 0; k < 4; ++k) {
                const int vv = mesh.F[t + (size_t)nT * k];
                if (owner[vv] == p) hGrad[(size_t)k * totalTets + inst] = (uint16_t)localIdx[vv];
            }
        }
        const int itemStart = (int)hP0.size();
        auto addBlock = [&](int p0, int L, bool isDiag, int row) {
            const int n = (int)cl.size();
            const int chunk = std::max(CHUNK, (n + MAXSEG - 1) / MAXSEG);
            const int segLen = std::max(1, (n + chunk - 1) / chunk);
            // a block's chunks must stay inside one wave
            const int posInWave = ((int)hP0.size() - itemStart) % 64;
            if (posInWave + segLen > 64)
                for (int pad = posInWave; pad < 64; ++pad) pushItem(-1, 0, 1, 0, false, 0, nullptr, nullptr);
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
        hItemPtr.push_back((int)hP0.size());
        for (int inst = hTetPtr[p]; inst < hTetPtr[p + 1]; ++inst) tetLocal[hTets[inst]] = -1;
    }
    hCPtr.push_back((int)hContrib.size());
    totalItems = (long long)hP0.size();
    totalContribs = (long long)hContrib.size();
    if (hContrib.empty()) hContrib.push_back(0);
    nodePtr.upload(hNodePtr, s);
    nodes.upload(hNodes, s);
    tetPtr.upload(hTetPtr, s);
    tets.upload(hTets, s);
    gradSlot.upload(hGrad, s);
    itemPtr.upload(hItemPtr, s);
    itemP0.upload(hP0, s);
    itemMeta.upload(hMeta, s);
    itemRow.upload(hRow, s);
    itemCPtr.upload(hCPtr, s);
    contrib.upload(hContrib, s);
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
    pv.itemP0 = itemP0.p;
    pv.itemMeta = itemMeta.p;
    pv.itemRow = itemRow.p;
    pv.itemCPtr = itemCPtr.p;
    pv.contrib = contrib.p;
    return pv;
}

size_t PatchPlan::ldsBytes() const
{
    const size_t tcap = (size_t)((maxTets + 63) / 64 * 64);
    return sizeof(double) * (NF * tcap + 3 * (size_t)maxNodes) + sizeof(int) * tcap;
}

void launch_assemble_patches(const ElemView& v, const PatchPlan& plan, int patchBegin, int patchEnd, double coef, int projectDBC,
    double* grad, double* a, hipStream_t s)
{
    const int n = patchEnd - patchBegin;
    if (n <= 0) return;
    const size_t lds = plan.ldsBytes();
    const int tcap = (plan.maxTets + 63) / 64 * 64;
    static size_t attrSet = 0;
    if (lds > 64 * 1024 && lds > attrSet) {
        HIP_CHECK(hipFuncSetAttribute((const void*)k_assemble_patch<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_CHECK(hipFuncSetAttribute((const void*)k_assemble_patch<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attrSet = lds;
    }
    const PatchView pv = plan.view();
    static const int probe = std::getenv("IPCGPU_ASM_PROBE") ? std::atoi(std::getenv("IPCGPU_ASM_PROBE")) : 0; // profiling only
    if (a)
        hipLaunchKernelGGL(k_assemble_patch<true>, dim3(n), dim3(BLOCK), lds, s, v, pv, patchBegin, tcap, plan.maxNodes, coef, projectDBC, grad,
            a, probe);
    else
        hipLaunchKernelGGL(k_assemble_patch<false>, dim3(n), dim3(BLOCK), lds, s, v, pv, patchBegin, tcap, plan.maxNodes, coef, projectDBC, grad,
            a, probe);
}

} // namespace ipcgpu
