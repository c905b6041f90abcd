// Atomic-free Newton assembly: owner-computes by node patch, gather by destination block.
//
// L2 atomics cap a tet-parallel scatter at ~30 G fp64 adds/s (90 adds per element, 8x the time of the
// arithmetic) and same-address LDS atomics are no better.  Here every workgroup owns a spatially compact patch
// of nodes, i.e. complete block rows of the symmetric-upper CSR (LinSysSolver.hpp:46-150):
//   phase 1  one lane per element touching the patch: F, P, SVD, clamped sigma-space matrices -> 36 doubles of
//            "generators" per element staged in LDS (halo elements are re-evaluated by the neighbouring patch)
//   phase 2  one lane per (owned destination block, <= 8 contributions): rebuilds each contributing 3x3 block
//            H_ac = U T_ac U^T from the staged generators, sums in registers, combines the chunks of a block by a
//            segmented reduction inside a 16-lane row (DPP row shifts), adds the lumped mass / Dirichlet identity
//            (Optimizer.cpp:3638-3668) and stores the block once.  Deterministic: fixed summation order.
//   gradient nodal forces accumulate in LDS (12 adds per element), inertia term added at the flush (:3438-3450)
// No memset, no second kernel, no global atomics: every CSR value and gradient entry is written exactly once.
#pragma once
#include <vector>
#include "common.h"
#include "nh_kernels.h"
#include <cstdint>

namespace ipcgpu {

class HipMesh;
class HipLinSysSolver;

struct PatchView {
    int nPatches;
    const int* nodePtr; // nPatches+1: owned nodes
    const int* nodes;
    const int* tetPtr; // nPatches+1: elements touching the patch (with halo)
    const int* tets;
    const uint16_t* gradSlot; // SoA [4][totalTets]: local index of the owned node or 0xFFFF
    long long totalTets;
    const int* itemPtr; // nPatches+1: work items of phase 2 (padded so that a block never straddles a 16-lane row)
    const int4* itemHdr; // x: CSR index of the block's first entry (row 0), -1 = padding; y: rowLen (16) | segLen (4) | segPos (4) |
                         // isDiag (1) | contributions of this chunk (7); z: row node of the block; w: first contribution
    const uint4* itemC4; // the chunk's first four contribution words (one 16-byte read instead of a pointer chase)
    const uint32_t* contrib; // tetLocal (16) | ka (2) | kc (2)
};

struct PatchPlan {
    int nPatches = 0;
    long long totalTets = 0, totalItems = 0, totalContribs = 0;
    int maxTets = 0, maxNodes = 0;
    double haloFactor = 0; // patch-tet instances / elements
    DevBuf<int> nodePtr, nodes, tetPtr, tets, itemPtr;
    DevBuf<int4> itemHdr;
    DevBuf<uint4> itemC4;
    DevBuf<uint32_t> contrib;
    DevBuf<uint16_t> gradSlot;
    bool valid = false;
    // The part of the plan that does not depend on the CSR pattern (node -> elements, Morton order, patches, gradient slots) is kept
    // on the host: when contact changes the pattern only the work items (CSR positions) are rebuilt, by a few threads.
    struct Topo {
        const void* mesh = nullptr;
        int nV = -1, nT = -1, tetCap = 0;
        unsigned version = 0;
        std::vector<int> vtPtr, vt, vtLoc, nodePtr, nodes, tetPtr, tets;
    } topo;

    PinnedBuf<int4> hHdrPin; // pinned staging of the work items (grow-only): they are rebuilt and shipped on every pattern change
    PinnedBuf<uint4> hC4Pin;
    PinnedBuf<uint32_t> hContribPin;

    void build(const HipMesh& mesh, const HipLinSysSolver& lin, hipStream_t s);
    PatchView view() const;
    size_t ldsBytes() const;
};

// grad / a may be null (Hessian-only or gradient-only).  patchBegin/End select a contiguous range of patches; patchList (device, patchEnd - patchBegin
// entries, patchBegin = 0) an arbitrary subset: the patches in which this rank owns CSR rows (owner-computes sharding, round 4).
void launch_assemble_patches(const ElemView& v, const PatchPlan& plan, int patchBegin, int patchEnd, double coef, int projectDBC,
    double* grad, double* a, hipStream_t s, const int* patchList = nullptr);

} // namespace ipcgpu
