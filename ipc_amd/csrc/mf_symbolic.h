// Host-side symbolic analysis for the GPU multifrontal Cholesky: fill-reducing nested-dissection
// ordering of the node graph (3x3 block structure of LinSysSolver.hpp:46-150), assembly tree,
// front index lists and the maps the numeric kernels consume.  This is the work the reference
// delegates to cholmod_analyze (CHOLMODSolver.cpp:123-128); rocSOLVER's csrrf path needs the same
// information handed to it (ordering + pattern of L), so it is produced here once per pattern.
#pragma once
#include <cstdint>
#include <vector>

namespace ipcgpu {

struct MfSymbolic {
    int n = 0; // scalar rows (3 * nn)
    int nn = 0; // nodes
    std::vector<int> newOf, oldOf; // node permutation: newOf[old] = new position
    int ns = 0; // fronts (supernodes)
    std::vector<int> firstNode; // ns+1, in new numbering; front s owns nodes [firstNode[s], firstNode[s+1])
    std::vector<int> idxPtr; // ns+1 into idx
    std::vector<int> idx; // per front: own nodes followed by the below-struct nodes (new numbering), ascending
    std::vector<int> parent, level;
    std::vector<int> childPtr, child; // children lists (ascending front id)
    std::vector<int> invPtr; // per front (as a child): offset into inv; length = #index nodes of its parent
    std::vector<int> inv; // parent-local node -> position in this child's struct list, or -1
    std::vector<int64_t> frontOff; // ns+1, offsets (in doubles) of the N x N column-major fronts
    std::vector<int64_t> wOff; // ns+1, offsets of the per-front solve work vectors (length N)
    std::vector<int> levelPtr, levelFronts; // fronts grouped by level (leaves = level 0)
    std::vector<int64_t> aDst; // per CSR entry of the user matrix: destination offset in the front buffer
    std::vector<int> aFront; // per CSR entry: the front that owns it (the front of its column in the permuted lower triangle)
    int64_t nnzL = 0;
    double flops = 0;
    int maxN = 0;

    int N(int s) const { return 3 * (idxPtr[s + 1] - idxPtr[s]); }
    int nc(int s) const { return 3 * (firstNode[s + 1] - firstNode[s]); }
};

// ia/ja: 0-based symmetric-upper CSR (scalar).  coords: optional nn x 3 row-major node coordinates used for
// geometric bisection (rest positions); when null the bisection direction is a BFS level structure.
// withEntryDestinations = false: aDst / aFront are left empty (the numeric phase computes them on the device from the pattern it is handed, MfNumeric::setup)
void mf_analyze(int n, const int* ia, const int* ja, const double* coords, int leafSize, MfSymbolic& out, bool withEntryDestinations = true);
void mf_entry_destinations(int n, const int* ia, const int* ja, MfSymbolic& sym);

// Multi-GPU: cut the assembly tree below its top separators.  The most expensive subtree that still has children is opened until there are at least `world`
// subtree roots; those go to the ranks greedily by factorisation cost (largest first, least-loaded rank), every front below a root inherits its rank,
// the opened fronts above the cut are shared (owner -1: every rank factorises them redundantly from exchanged update matrices).  Returns the share of the
// factorisation flops above the cut.  world <= 1: every front owned by rank 0.
double mf_assign_owners(const MfSymbolic& sym, int world, std::vector<int>& owner);
// Round 5: the fronts above the cut are no longer repeated by every rank -- each is EXECUTED by one rank, the executor of the child with the most expensive
// subtree (ties: the first child), so the update matrix of that child never travels and the other children's go point to point to the one rank that
// needs them.  exec[s] = owner[s] below the cut.  group[s] (bit r set: rank r executes a front of the subtree of s; world <= 64) tells who needs the
// solution entries of s in the backward sweep.
void mf_assign_executors(const MfSymbolic& sym, const std::vector<int>& owner, std::vector<int>& exec, std::vector<unsigned long long>& group);
// What rank `rank` sends and receives, level by level (host logic of MfNumeric's exchanges, shared with the CPU tests of the protocol):
//   send / recv: fronts of the level whose parent another rank executes -- `off` = offset of the packed update matrix (m (m + 1) / 2 doubles, m = N - nc)
//                in the level's staging area, `offW` = offset of the update vector (m doubles) in the vector area behind it; every rank computes the
//                same layout.  After the level's factorisation the matrices travel, after its forward sweep the vectors.
//   xs:          after the level's backward sweep: the solution entries (nc doubles at 3 firstNode) of a front above the cut go from its executor
//                to every other rank of its group.
struct MfExchangeItem {
    int front;
    long long off;
    int offW;
    int peer;
};
struct MfExchangeLevel {
    std::vector<MfExchangeItem> send, recv, xsSend, xsRecv;
    long long count = 0; // doubles of the matrix area
    int countW = 0; // doubles of the vector area
};
void mf_exchange_plan(const MfSymbolic& sym, const std::vector<int>& owner, const std::vector<int>& exec, const std::vector<unsigned long long>& group, int rank,
    int world, std::vector<MfExchangeLevel>& plan);

// scalar CSR pattern of L (lower triangle incl. diagonal, rows sorted) in the permuted ordering plus the
// scalar permutation pivQ (new -> old) -- what rocsolver_dcsrrf_analysis expects as T and pivQ.
void mf_L_pattern_csr(const MfSymbolic& sym, std::vector<int>& ptrT, std::vector<int>& indT, std::vector<int>& pivQ);

} // namespace ipcgpu
