// Cyclic Jacobi eigen-iteration of a small symmetric matrix (M = 3, 6, 9) with the iteration matrices in REGISTERS: the PSD projection of a
// contact stencil's barrier Hessian, carried out in the 3 (nn - 1)-dimensional complement of the rigid translations.
//   IglUtils::makePD (IglUtils.hpp:119-137) on the 3 nn x 3 nn block of a stencil of nn = 2..4 nodes,
//   call sites SelfCollisionHandler.cpp:418-561, 3039-3201
//
// One lane per stencil.  The M (M - 1) / 2 rotations of a sweep are that many template instances: every index is a compile-time constant, the
// M (M + 1) / 2 entries of the symmetric iterate and the M^2 of the accumulated rotations are plain scalars (M = 9: 252 VGPRs; gfx950 gives a
// wave 512), and a rotation is ~100 independent fp64 instructions.  (Rounds 1-2 walked two LDS-resident matrices with run-time indices; rounds 3-5
// embedded 2- and 3-node stencils in the 9 x 9 frame.  Round 6: the stencil kind is a template parameter of the Hessian kernel --
// stencil_hessian_device.h -- so a point-point stencil iterates on 3 x 3 and a point-edge stencil on 6 x 6.)
//
// The order of the rotations and the stopping rule (off-diagonal norm <= 1e-14 of the diagonal norm, per stencil) do not depend on which
// other stencils share the wave: a wave-wide vote only skips work that NO lane needs.
//
// The file compiles for the host as well (tests/test_jacobi9.py builds it with g++ and checks it against numpy's eigh).
#pragma once
#include <cmath>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define J9_HD __device__ __forceinline__
#define J9_WAVE_ANY(x) (__builtin_amdgcn_ballot_w64(x) != 0ull)
#else
#define J9_HD inline
#define J9_WAVE_ANY(x) (x)
#endif
// the translation unit that holds the contact typing is compiled with -ffp-contract=off (exact comparisons there); nothing here
// compares exactly, so multiply-adds may fuse
#ifdef __clang__
#define J9_CONTRACT _Pragma("clang fp contract(fast)")
#else
#define J9_CONTRACT
#endif

namespace ipcgpu {
namespace j9 {

// upper-triangle accessor of the symmetric iterate: only these M (M + 1) / 2 slots of the M x M array are ever named, the rest never exist
template <int M>
constexpr int us(int i, int j) { return i <= j ? i * M + j : j * M + i; }

template <int M, int P, int Q>
J9_HD void rotate(double (&W)[M * M], double (&V)[M * M], bool live)
{
    J9_CONTRACT
    const double apq = W[us<M>(P, Q)];
    const bool rot = live && apq != 0.0;
    if (!J9_WAVE_ANY(rot)) return;
    const double app = W[us<M>(P, P)], aqq = W[us<M>(Q, Q)];
    // tan of the rotation angle: t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)), theta = (aqq - app) / (2 apq), written without the quotient
    // theta: t = sgn(a) b / (|a| + sqrt(a^2 + b^2)) with a = (aqq - app) / 2, b = apq.  One square root and one division instead of two of
    // each (an IEEE fp64 division or square root is ~35 instructions here: they were half of a rotation).  On the device both come from the
    // hardware estimates with Newton steps: t only steers the iteration (an error in it costs convergence speed, nothing else), while
    // c = rsqrt(1 + t^2) is refined to full precision for the t actually used -- c^2 + s^2 = 1 is what keeps the rotation orthogonal.
    const double a = 0.5 * (aqq - app), b = rot ? apq : 1.0;
    const double h2 = a * a + b * b;
#ifdef __HIPCC__
    double rh = __builtin_amdgcn_rsq(h2);
    rh = rh * (1.5 - 0.5 * h2 * rh * rh);
    const double den = fabs(a) + h2 * rh;
    double rd = __builtin_amdgcn_rcp(den);
    rd = rd * (2.0 - den * rd);
    rd = rd * (2.0 - den * rd);
    const double tt = (a >= 0 ? b : -b) * rd;
    const double u = tt * tt + 1.0;
    double cc = __builtin_amdgcn_rsq(u);
    cc = cc * (1.5 - 0.5 * u * cc * cc);
    cc = cc * (1.5 - 0.5 * u * cc * cc);
    cc = cc * (1.5 - 0.5 * u * cc * cc);
#else
    const double tt = (a >= 0 ? b : -b) / (fabs(a) + sqrt(h2));
    const double cc = 1.0 / sqrt(tt * tt + 1.0);
#endif
    const double t = rot ? tt : 0.0, c = rot ? cc : 1.0, s = rot ? tt * cc : 0.0;
#pragma unroll
    for (int k = 0; k < M; ++k) {
        if (k != P && k != Q) {
            const double wp = W[us<M>(k, P)], wq = W[us<M>(k, Q)];
            const double np = c * wp - s * wq, nq = s * wp + c * wq;
            W[us<M>(k, P)] = rot ? np : wp;
            W[us<M>(k, Q)] = rot ? nq : wq;
        }
        const double vp = V[k + M * P], vq = V[k + M * Q];
        const double mp = c * vp - s * vq, mq = s * vp + c * vq;
        V[k + M * P] = rot ? mp : vp;
        V[k + M * Q] = rot ? mq : vq;
    }
    W[us<M>(P, P)] = app - t * apq;
    W[us<M>(Q, Q)] = aqq + t * apq;
    W[us<M>(P, Q)] = rot ? 0.0 : apq;
}

// the rotations (P, Q), (P, Q + 1), ... of one cyclic sweep, row by row
template <int M, int P, int Q>
J9_HD void sweep_from(double (&W)[M * M], double (&V)[M * M], bool live)
{
    if constexpr (P < M - 1) {
        rotate<M, P, Q>(W, V, live);
        sweep_from<M, (Q + 1 < M) ? P : P + 1, (Q + 1 < M) ? Q + 1 : P + 2>(W, V, live);
    }
}

// W: symmetric M x M (upper triangle, us<M>) -> its eigenvalues on the diagonal; V: the accumulated rotations (columns = eigenvectors).
// Returns the number of sweeps this lane took.
template <int M>
J9_HD int jacobi_sweeps(double (&W)[M * M], double (&V)[M * M])
{
    J9_CONTRACT
#pragma unroll
    for (int e = 0; e < M * M; ++e) V[e] = (e % M == e / M) ? 1.0 : 0.0;
    int sweeps = 0;
    bool live = true;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dg = 0.0;
#pragma unroll
        for (int j = 0; j < M; ++j)
#pragma unroll
            for (int i = 0; i <= j; ++i) {
                const double v = W[us<M>(i, j)];
                if (i != j) off += 2.0 * v * v;
                else dg += v * v;
            }
        live = live && !(off <= 1e-28 * dg || off == 0.0);
        if (!J9_WAVE_ANY(live)) break;
        sweeps += live ? 1 : 0;
        sweep_from<M, 0, 1>(W, V, live);
    }
    return sweeps;
}

} // namespace j9
} // namespace ipcgpu
