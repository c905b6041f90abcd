// Launchers of the per-tetrahedron neo-Hookean kernels and the nodal helper kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace ipcgpu {

// Device view of the mesh / CSR map handed to the element kernels.  All pointers are HBM.
struct ElemView {
    int nV, nT;
    int tetBegin, tetEnd; // element shard of this rank
    int energyType; // Config `energy`: 0 NH (neo-Hookean, element-inversion safeguard on), 1 FCR (fixed corotated) (Config.cpp:23-24)
    const double* x; // positions, xyz interleaved (24 B per node: one gather touches one or two lines)
    const double* xTilde; // same layout
    const double* mass; // nV
    const int* dbc; // nV, DirichletBCType
    const int4* tet; // nT vertex ids
    const double* A; // restTriInv, SoA: A[k * nT + t], k = i + 3 j (column-major entry k of tet t)
    const double* vol; // triArea (rest volume)
    const double* mu;
    const double* lam;
    // CSR map (valid after set_pattern)
    const int* rowBase; // nV: ia[3 v]
    const int* rowLen; // nV: ia[3 v + 1] - ia[3 v]
    const int* edgeP0; // SoA [6][nT]: first CSR slot of the 3x3 block of local edge e in row 3*min(va,vb)
};

// a[] / grad[] initialisation: mass or identity on the diagonal, inertia term of the gradient.
// (computePrecondMtr mass/DBC part, Optimizer.cpp:3638-3668; computeGradient inertia, :3438-3450)
void launch_node_init(const ElemView& v, int projectDBC, bool ownerRank, double* a, double* grad, hipStream_t s);

// fused elastic gradient (+ optional projected Hessian into CSR) -- the Newton assembly kernel
void launch_assemble(const ElemView& v, double coef, int projectDBC, double* grad /*nullable*/,
    double* a /*nullable*/, hipStream_t s);
const char* assemble_kernel_name();

// energy: partial[0..nBlocks) then reduced into *out (device) deterministically.
// E = coef * sum vol psi + (withInertia ? sum 1/2 m |x - xTilde|^2 : 0)
void launch_energy(const ElemView& v, double coef, bool withInertia, bool ownerRank, double* partial, int partialCap,
    double* out, hipStream_t s);
void launch_energy_per_elem(const ElemView& v, double* perElem, hipStream_t s);

// inversion: flag[0] |= any det < 0 ; step bound: per-shard min written to *outMin (must be preset to +inf bits)
void launch_check_inversion(const ElemView& v, int* flag, hipStream_t s);
// tMax: roots from there on do not matter to the caller (elements that provably have none below it skip the closed form)
void launch_inversion_step(const ElemView& v, const double* p, double slackness, double tMax, double* outMin, hipStream_t s);

// nodal vector helpers
void launch_step_forward(int n3, const double* x0, const double* p, double alpha, double* x, hipStream_t s);
// alpha = 1, or the inversion filter's minimum when it applies, decided on the device; x = x0 + alpha p
void launch_trial_step(int n3, const double* x0, const double* p, const double* filterMin, bool useFilter, double* alphaOut, double* x, hipStream_t s);
void launch_max_abs(int n, const double* v, double* out /*preset 0*/, hipStream_t s);
// round 4: the small launches of a contact-free iteration's tail, fused (x -> x0 + alpha + step; scalar resets; the two read-backs)
void launch_trial_step_fused(int n3, double* x, double* x0, const double* p, const double* filterMin, bool useFilter, double* alphaOut, hipStream_t s);
void launch_iter_reset(double* scalar, int* flag, hipStream_t s);
void launch_publish2(const void* a, void* da, int na, const void* b, void* db, int nb, hipStream_t s);
void launch_fill(double* p, size_t n, double v, hipStream_t s);
void launch_negate(int n, const double* in, double* out, hipStream_t s);
// copies nWords 4-byte words from device memory into mapped pinned host memory (a ~24 us blit per scalar read-back otherwise)
void launch_publish(const void* src_dev, void* dst_mapped, int nWords, hipStream_t s);
void launch_colmajor_to_aos(int nV, const double* src, double* dst, hipStream_t s);
void launch_aos_to_colmajor(int nV, const double* src, double* dst, hipStream_t s);
// symmetric-upper CSR times vector and diagonal preconditioner (LinSysSolver.hpp:238-253, 411-420)
void launch_csr_symv(int nRows, const int* ia, const int* ja, const double* a, const double* x, double* y, hipStream_t s);
// a[k] = mask && mask[k] ? setVal[k] + delta[k] : a[k] + delta[k]  (host-side addCoeff / setCoeff of an adapter, flushed in one pass)
void launch_apply_host_updates(long long nnz, const double* delta, const unsigned char* mask, const double* setVal, double* a, hipStream_t s);
void launch_precond_diag(int nRows, const int* ia, const double* a, const double* in, double* out, hipStream_t s);
// BE update (Optimizer.cpp:570-580, 1236-1257): dxElastic = x - xTilde; acc = (vel_new - vel) / dt; vel = (x - xPrev) / dt;
// xPrev = x; xTilde = xPrev + dt vel + dt^2 g (DBC: xPrev)
void launch_be_update(int nV, const int* dbc, const double* x, double* xPrev, double* vel, double* acc, double* dxElastic, double* xTilde,
    double dt, double gx, double gy, double gz, hipStream_t s);
// Newmark update (Optimizer.cpp:582-590, 1259-1277)
void launch_nm_update(int nV, const int* dbc, const double* x, double* xPrev, double* vel, double* acc, double* dxElastic, double* xTilde,
    double dt, double beta, double gamma, double gx, double gy, double gz, hipStream_t s);
// scripted Dirichlet groups (Mesh::DirichletBCs / scripted component velocities, AnimScripter.cpp:1413-1462)
struct DbcMotion {
    double R[9]; // row-major Rx Ry Rz of angVel dt
    double c[3]; // centre of the group's current bounding box
    double linDt[3];
};
void launch_gather3(int n, const int* ids, const double* x, double* out, hipStream_t s);
void launch_target_positions(int n, const int* ids, const double* x, const double* p, double* pos, double* pOut, hipStream_t s);
void launch_dbc_motion(int n, const int* ids, const DbcMotion& m, const double* x, double* p, hipStream_t s);
void launch_dbc_targets(int n, const int* ids, const double* target_3n, const double* x, double* p, hipStream_t s);
// Neumann boundary conditions (Mesh::NeumannBCs; Optimizer.cpp:3241-3250, 3452-3461): dtSqA3 = dt^2 * acceleration
void launch_nbc_gradient(int n, const int* ids, const int* dbc, const double* mass, const double* dtSqA3, double* g, hipStream_t s);
void launch_nbc_energy(int n, const int* ids, const int* dbc, const double* mass, const double* x, const double* dtSqA3, double* out, hipStream_t s);
// augmented-Lagrangian Dirichlet fallback (AnimScripter.cpp:2303-2346; Optimizer.cpp:3402-3404, 3542-3544, 3711-3713)
struct MdbcView {
    int n; // target positions
    const int* ids;
    const double* pos; // 3 n
    double* lam; // 3 n
    const double* mass; // nV
};
void launch_clear_projected(int nV, const int* dbc, int projectDBC, double* g, hipStream_t s);
void launch_keep_mine3(int nV, const unsigned char* mine, double* g, hipStream_t s); // owner-computes sharding: zero what other ranks contribute
void launch_keep_mine_rows(int nRows, const unsigned char* mine, const int* ia, double* a, hipStream_t s);
// lagged damping (Optimizer.cpp:3381-3400, 3519-3540, 3707-3709): displacement of the step (mode 0: rows of every Dirichlet node
// cleared, 1: of the projected ones), the diagonal fix-up of the damping matrix, y += alpha x, out = scale * x . y (one workgroup)
void launch_damp_dx(int nV, const int* dbc, int mode, int projectDBC, const double* x, const double* xPrev, double* dx, hipStream_t s);
void launch_damp_clear_diag(int nV, const int* dbc, const int* ia, double* d, hipStream_t s);
void launch_axpy(long long n, double alpha, const double* x, double* y, hipStream_t s);
void launch_dot_scaled(int n, const double* x, const double* y, double scale, double* out, hipStream_t s);
// out2[0] = g . e, out2[1] = g . g (many workgroups, fixed order; partial: 2 x min(128, partialCap / 2) doubles of scratch)
void launch_dot2(int n, const double* g, const double* e, double* partial, int partialCap, double* out2, hipStream_t s);
void launch_mdbc_reduce(const MdbcView& m, const double* x, double rho, int mode /*0 energy, 1 |x - target|^2*/, double* out, hipStream_t s);
void launch_mdbc_gradient(const MdbcView& m, const double* x, double rho, double* g, hipStream_t s);
void launch_mdbc_hessian(const MdbcView& m, const int* ia, double rho, double* a, hipStream_t s);
void launch_mdbc_lambda(const MdbcView& m, const double* x, double rho, hipStream_t s);
// twist handles: rotate listed vertices about the x axis through c by their angle (AnimScripter.cpp:1674-1684)
// p = dbc ? 0 : dt vel + cgDtSqG + ce dx  (initX options 1-4, Optimizer.cpp:936-1080)
void launch_warm_dir(int nV, const int* dbc, const double* vel, const double* dx, double dt, const double* cgDtSqG3, double ce, double* p,
    hipStream_t s);
void launch_twist_dir(int nH, const int* ids, const double* ang, double cy, double cz, const double* x, double* p, hipStream_t s);

} // namespace ipcgpu
