// Shared by the translation units of the numeric phase (mf_numeric.hip: set-up + factorisation, mf_sweeps.hip: triangular sweeps, mf_exchange.hip: what crosses
// ranks): tuning constants, the device views of the assembly tree and of the explicit inverses.  Internal linkage on purpose (every unit compiles its own copy).
#pragma once
#include "mf_numeric.h"

namespace ipcgpu {

namespace {


constexpr int NB = 32;
constexpr int LDP = NB + 1; // padded leading dimension of 32x32 blocks in LDS
constexpr int LDX = NB + 1; // same for the inverse of a pivot block
constexpr int WG = 256;
constexpr int WGB = WG; // big-front step: three row waves + one pivot wave, one per SIMD
constexpr int ROW_WAVES_B = 3; // row waves per role-B workgroup (1 was measured slower: 3x the workgroups, each repeating the pivot work)
#ifndef MF_ROWS_MT
#define MF_ROWS_MT 2
#endif
#ifndef MF_SCHUR_OCC
#define MF_SCHUR_OCC 1 // the same for k_big_schur / k_big_schur64 (the levels without the fused extend-add)
#endif
#ifndef MF_STEP_OCC
#define MF_STEP_OCC 3 // waves per SIMD the step kernel is compiled for: 130 registers, no scratch (1 = no constraint: 180 registers, 2 waves; +0.7 % at 45 K nodes,
                      // profiles/r05_solver_ab_xcd_occupancy.txt)
#endif
constexpr int MT_B = MF_ROWS_MT; // 16-row tiles per row wave of a role-B workgroup
constexpr int ROWS_B = 16 * MT_B * ROW_WAVES_B; // panel rows per role-B workgroup
constexpr int WGT = 512; // workgroup of the big-front triangular sweeps

constexpr int TS = 64; // trailing-update tile
constexpr int XCDS = 8; // accelerator complex dies of an MI355X: workgroup b of a launch is observed to run on XCD b % 8
constexpr int MV_ROWS = 32; // rows per workgroup of the forward matrix-vector kernels (k_big_fwd_rect, k_xinv_fwd): 32 rows x 8 column groups
constexpr int FD_STRIDE_EA = 64; // packed front descriptors (same layout as the fused kernel's, see k_front_fused)
constexpr int FUSED_MAX_KIDS_EA = 8;
typedef double f64x4 __attribute__((ext_vector_type(4)));

struct TreeView {
    const long long* frontOff;
    const int* idxPtr;
    const int* firstNode;
    const int* childPtr;
    const int* child;
    const int* invPtr;
    const int* inv;
    const int* idx;
    const long long* dinvOff; // per front: first 32x32 inverse block (in blocks)
};

__device__ __forceinline__ int frontN(const TreeView& tv, int s) { return 3 * (tv.idxPtr[s + 1] - tv.idxPtr[s]); }
__device__ __forceinline__ int frontNc(const TreeView& tv, int s) { return 3 * (tv.firstNode[s + 1] - tv.firstNode[s]); }

// ---- explicit inverses of the factor triangles of the widest fronts (see the k_xinv_* kernels further down for what they are used for)
struct XinvView {
    const long long* xOff; // per front: offset of X (and of the scratch T) in their buffers, -1 when the front has none
    double* X;
    double* T;
};

} // namespace

} // namespace ipcgpu
