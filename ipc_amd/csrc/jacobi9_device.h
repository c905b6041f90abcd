// PSD projection of a contact stencil's Hessian block with the iteration matrices in REGISTERS.
//   IglUtils::makePD (IglUtils.hpp:119-137) on the 3 nn x 3 nn block of a stencil of nn = 2..4 nodes,
//   call sites SelfCollisionHandler.cpp:418-561, 3039-3201
//
// One lane per stencil, as before -- but the two 9 x 9 matrices of the cyclic Jacobi iteration used to be walked with run-time row /
// column indices, which forces them out of the register file: in private memory every access is a scratch round trip, in LDS (the
// previous version, 1.3 KB per stencil) a workgroup is 32 lanes and a CU holds three of them, each rotation a load -> wait ->
// compute -> store chain with nothing to hide it behind.  Here the 36 rotations of a sweep are 36 template instances: every index
// is a compile-time constant, the 45 entries of the symmetric iterate and the 81 of the accumulated rotations are plain scalars
// (252 VGPRs; gfx950 gives a wave 512), and a rotation is ~100 independent fp64 instructions.  Blocks of 2- and 3-node stencils are
// embedded in the 9 x 9 frame: their zero rows make the extra rotations identities (apq == 0), skipped per wave when no lane needs them.
//
// The order of the rotations, the stopping rule (off-diagonal norm <= 1e-14 of the diagonal norm, per
// stencil) and the reduction to the complement of the three rigid translations are those of make_pd_stencil (contact_device.h), so a
// stencil's result does not depend on which other stencils share its wave.
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

constexpr int M = 9;
// upper-triangle accessor of the symmetric iterate: only these 45 slots of the 81-array are ever named, the rest never exist
constexpr int us(int i, int j) { return i <= j ? i * M + j : j * M + i; }

template <int P, int Q>
J9_HD void rotate(double (&W)[81], double (&V)[81], bool live)
{
    J9_CONTRACT
    const double apq = W[us(P, Q)];
    const bool rot = live && apq != 0.0;
    if (!J9_WAVE_ANY(rot)) return;
    const double app = W[us(P, P)], aqq = W[us(Q, Q)];
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
            const double wp = W[us(k, P)], wq = W[us(k, Q)];
            const double np = c * wp - s * wq, nq = s * wp + c * wq;
            W[us(k, P)] = rot ? np : wp;
            W[us(k, Q)] = rot ? nq : wq;
        }
        const double vp = V[k + M * P], vq = V[k + M * Q];
        const double mp = c * vp - s * vq, mq = s * vp + c * vq;
        V[k + M * P] = rot ? mp : vp;
        V[k + M * Q] = rot ? mq : vq;
    }
    W[us(P, P)] = app - t * apq;
    W[us(Q, Q)] = aqq + t * apq;
    W[us(P, Q)] = rot ? 0.0 : apq;
}

template <int P, int Q>
struct Sweep {
    static J9_HD void run(double (&W)[81], double (&V)[81], bool live)
    {
        rotate<P, Q>(W, V, live);
        Sweep<(Q + 1 < M) ? P : P + 1, (Q + 1 < M) ? Q + 1 : P + 2>::run(W, V, live);
    }
};
template <>
struct Sweep<M - 1, M> {
    static J9_HD void run(double (&)[81], double (&)[81], bool) {}
};

// A: the 12 x 12 block (leading dimension 12, rows / columns >= 3 nn zero), overwritten by its projection when it has a negative
// eigenvalue and left untouched otherwise.  Returns the number of sweeps this stencil took.
J9_HD int make_pd_stencil_reg(int nn, double* A)
{
    J9_CONTRACT
    // Helmert basis of the complement of (1, ..., 1): h[a][k], a < nn - 1
    double h[3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double sc = 1.0 / sqrt((double)((a + 1) * (a + 2)));
#pragma unroll
        for (int k = 0; k < 4; ++k) h[a][k] = (a + 1 < nn && k < nn) ? (k <= a ? sc : (k == a + 1 ? -(a + 1) * sc : 0.0)) : 0.0;
    }
    // C = R^T A R with R = H (x) I3, both triangles (they differ by rounding) averaged into the upper one
    double W[81], V[81];
    {
        double C[81];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double d[12]; // row (3 a + i) of R^T A
#pragma unroll
                for (int c = 0; c < 12; ++c) {
                    double v = 0.0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v += h[a][k] * A[(3 * k + i) + 12 * c];
                    d[c] = v;
                }
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        double v = 0.0;
#pragma unroll
                        for (int l = 0; l < 4; ++l) v += h[b][l] * d[3 * l + j];
                        C[(3 * a + i) + M * (3 * b + j)] = v;
                    }
            }
#pragma unroll
        for (int j = 0; j < M; ++j)
#pragma unroll
            for (int i = 0; i <= j; ++i) W[us(i, j)] = (i == j) ? C[i + M * i] : 0.5 * (C[i + M * j] + C[j + M * i]);
    }
#pragma unroll
    for (int e = 0; e < 81; ++e) V[e] = (e % M == e / M) ? 1.0 : 0.0;
    int sweeps = 0;
    bool live = true;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dg = 0.0;
#pragma unroll
        for (int j = 0; j < M; ++j)
#pragma unroll
            for (int i = 0; i <= j; ++i) {
                const double v = W[us(i, j)];
                if (i != j) off += 2.0 * v * v;
                else dg += v * v;
            }
        live = live && !(off <= 1e-28 * dg || off == 0.0);
        if (!J9_WAVE_ANY(live)) break;
        sweeps += live ? 1 : 0;
        Sweep<0, 1>::run(W, V, live);
    }
    const int m = 3 * (nn - 1);
    double ev[M];
    double wmin = W[us(0, 0)];
#pragma unroll
    for (int i = 0; i < M; ++i) {
        ev[i] = (i < m) ? W[us(i, i)] : 0.0;
        wmin = fmin(wmin, ev[i]);
    }
    if (!J9_WAVE_ANY(wmin < 0.0)) return sweeps;
    // C+ = V max(ev, 0) V^T (upper triangle, into W), then A = R C+ R^T
#pragma unroll
    for (int j = 0; j < M; ++j)
#pragma unroll
        for (int i = 0; i <= j; ++i) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < M; ++k) v += (ev[k] > 0.0) ? V[i + M * k] * ev[k] * V[j + M * k] : 0.0;
            W[us(i, j)] = v;
        }
    if (wmin >= 0.0) return sweeps; // this stencil's block was positive semi-definite already: untouched
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double d[M]; // column (3 l + j) of C+ R^T: d[r] = sum_b C+[r, 3 b + j] h[b][l]
#pragma unroll
            for (int r = 0; r < M; ++r) {
                double v = 0.0;
#pragma unroll
                for (int b = 0; b < 3; ++b) v += W[us(r, 3 * b + j)] * h[b][l];
                d[r] = v;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    double v = 0.0;
#pragma unroll
                    for (int a = 0; a < 3; ++a) v += h[a][k] * d[3 * a + i];
                    A[(3 * k + i) + 12 * (3 * l + j)] = v;
                }
        }
    return sweeps;
}

} // namespace j9
} // namespace ipcgpu
