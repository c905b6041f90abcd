// Per-tetrahedron neo-Hookean kernels for gfx950 (CDNA4), fp64.
//
// What the reference does per element on the CPU (tbb::parallel_for bodies):
//   energy   Energy.cpp:195-242  + NeoHookeanEnergy.cpp:55-69
//   gradient Energy.cpp:334-366  + NeoHookeanEnergy.cpp:138-153 + IglUtils.cpp:635-669
//   Hessian  Energy.cpp:368-408, 448-562 + NeoHookeanEnergy.cpp:71-136 + IglUtils.hpp:337-432
//   scatter  Energy.cpp:270-288, 317-327 -> IglUtils.hpp:39-116 -> LinSysSolver.hpp:402-410
//   step     get_feasible_steps.cpp:75-172
// Here one lane owns one tetrahedron.  Element data is SoA so a wave reads 512 contiguous
// bytes per field; nodal positions are 24-byte records gathered through L2.  The 9x9
// dP/dF of the reference (21-term sums per entry, Energy.cpp:552) is never formed: with
// beta_a = V^T (dN_a/dX) the 3x3 node-pair block is  H_ac = U T_ac U^T  where T_ac collects
// the sigma-space A block (3x3, PSD-clamped) and the three 2x2 B blocks (makePD2d-clamped)
// -- the same numbers, ~2.5 kflop instead of ~7 kflop per element.
#include "nh_device.h"
#include <algorithm>
#include <cfloat>
#include <cstdlib>

namespace ipcgpu {

namespace {

constexpr int BLOCK = 256;

using namespace dev;

// ------------------------------------------------------------------------------------------------
// Tet-parallel assembly with hardware fp64 atomics (global_atomic_add_f64).  This is the simple path: it serves
// ipcgpu_elastic_gradient / ipcgpu_elastic_hessian_add and cross-checks the patch kernel in the tests.  The
// Newton loop uses the atomic-free patch kernel (patch_assembly.hip): L2 atomics cap this one at ~30 G adds/s.
struct GlobalAtomicSink {
    const ElemView& v;
    int t;
    int vid[4];
    double* gptr;
    double* a;
    __device__ __forceinline__ void grad(int k, int i, double g) { atomicAdd(&gptr[3 * (size_t)vid[k] + i], g); }
    __device__ __forceinline__ bool wantPair(int, int, int) const { return true; }
    __device__ __forceinline__ void diagBlock(int ka, const double H[3][3])
    {
        const int base = v.rowBase[vid[ka]], Lr = v.rowLen[vid[ka]];
        atomicAdd(&a[base + 0], H[0][0]);
        atomicAdd(&a[base + 1], H[0][1]);
        atomicAdd(&a[base + 2], H[0][2]);
        atomicAdd(&a[base + Lr + 0], H[1][1]);
        atomicAdd(&a[base + Lr + 1], H[1][2]);
        atomicAdd(&a[base + 2 * Lr - 1], H[2][2]);
    }
    __device__ __forceinline__ void offBlock(int e, int ka, int kc, bool aFirst, const double H[3][3])
    {
        // block row = smaller global id; H has rows of node ka and columns of node kc
        const int p0 = v.edgeP0[(size_t)e * v.nT + t];
        const int Lr = v.rowLen[vid[aFirst ? ka : kc]];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int rowOff = (r == 0) ? 0 : (r == 1 ? (Lr - 1) : (2 * Lr - 3));
#pragma unroll
            for (int c = 0; c < 3; ++c) atomicAdd(&a[p0 + rowOff + c], aFirst ? H[r][c] : H[c][r]);
        }
    }
};

template <bool HESS>
__global__ __launch_bounds__(BLOCK) void k_assemble(ElemView v, double coef, int projectDBC, double* __restrict__ grad,
    double* __restrict__ a)
{
    const int t = v.tetBegin + blockIdx.x * BLOCK + threadIdx.x;
    if (t >= v.tetEnd) return;
    const int4 tv = v.tet[t];
    GlobalAtomicSink sink{ v, t, { tv.x, tv.y, tv.z, tv.w }, grad, a };
    assemble_element<HESS>(v, t, coef, projectDBC, grad != nullptr, sink);
}

__global__ __launch_bounds__(BLOCK) void k_node_init(ElemView v, int projectDBC, int owner, double* __restrict__ a,
    double* __restrict__ grad)
{
    const int n = blockIdx.x * BLOCK + threadIdx.x;
    if (n >= v.nV) return;
    const int type = v.dbc[n];
    const bool proj = projected_dbc(type, projectDBC);
    if (grad) {
        double g[3] = { 0, 0, 0 };
        if (!proj && owner) {
            const double m = v.mass[n];
#pragma unroll
            for (int i = 0; i < 3; ++i) g[i] = m * (v.x[3 * (size_t)n + i] - v.xTilde[3 * (size_t)n + i]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) grad[3 * (size_t)n + i] = g[i];
    }
    if (a && owner) {
        const int base = v.rowBase[n], Lr = v.rowLen[n];
        const double d = proj ? 1.0 : v.mass[n];
        a[base] = d;
        a[base + Lr] = d;
        a[base + 2 * Lr - 1] = d;
    }
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double x, double* sm)
{
    // fixed-order tree: deterministic
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sm[wv] = x;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        for (int i = 0; i < BLOCK / 64; ++i) r += sm[i];
    }
    return r;
}

__device__ __forceinline__ double nh_psi(const double F[9], double mu, double lam)
{
    if (mu == 0.0 && lam == 0.0) return 0.0;
    double I1 = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) I1 += F[i] * F[i];
    const double lJ = log(det3(F));
    return mu / 2.0 * (I1 - 3.0) - (mu - lam / 2.0 * lJ) * lJ; // NeoHookeanEnergy.cpp:64-68 with sum sigma^2 = |F|^2, prod sigma = det F
}

// strain energy density of the configured material
__device__ __forceinline__ double elem_psi(const ElemView& v, const double F[9], double mu, double lam)
{
    if (v.energyType != 1) return nh_psi(F, mu, lam);
    if (mu == 0.0 && lam == 0.0) return 0.0;
    double U[9], s[3], V[9];
    svd3(F, U, s, V); // Energy.cpp:195-242 (computeEnergyValBySVD)
    return fcr_psi(s, mu, lam);
}

__global__ __launch_bounds__(BLOCK) void k_energy(ElemView v, double coef, int withInertia, int owner, double* __restrict__ partial)
{
    __shared__ double sm[BLOCK / 64];
    const int gid = blockIdx.x * BLOCK + threadIdx.x;
    double e = 0.0;
    const int t = v.tetBegin + gid;
    if (t < v.tetEnd) {
        const int4 tv = v.tet[t];
        double A[9], F[9];
        load_A(v, t, A);
        deformation_gradient(ld3(v.x, tv.x), ld3(v.x, tv.y), ld3(v.x, tv.z), ld3(v.x, tv.w), A, F);
        e = coef * v.vol[t] * elem_psi(v, F, v.mu[t], v.lam[t]);
    }
    if (withInertia && owner && gid < v.nV) {
        double d2 = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double d = v.x[3 * (size_t)gid + i] - v.xTilde[3 * (size_t)gid + i];
            d2 += d * d;
        }
        e += d2 * v.mass[gid] / 2.0; // Optimizer.cpp:3234
    }
    double r = block_sum(e, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

__global__ __launch_bounds__(BLOCK) void k_reduce_partials(const double* __restrict__ partial, int n, double* __restrict__ out)
{
    __shared__ double sm[BLOCK / 64];
    double x = 0.0;
    for (int i = threadIdx.x; i < n; i += BLOCK) x += partial[i];
    double r = block_sum(x, sm);
    if (threadIdx.x == 0) out[0] = r;
}

__global__ __launch_bounds__(BLOCK) void k_energy_per_elem(ElemView v, double* __restrict__ out)
{
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= v.nT) return;
    const int4 tv = v.tet[t];
    double A[9], F[9];
    load_A(v, t, A);
    deformation_gradient(ld3(v.x, tv.x), ld3(v.x, tv.y), ld3(v.x, tv.z), ld3(v.x, tv.w), A, F);
    out[t] = v.vol[t] * elem_psi(v, F, v.mu[t], v.lam[t]);
}

__global__ __launch_bounds__(BLOCK) void k_check_inversion(ElemView v, int* flag)
{
    const int t = v.tetBegin + blockIdx.x * BLOCK + threadIdx.x;
    if (t >= v.tetEnd) return;
    if (!(v.mu[t] != 0.0 && v.lam[t] != 0.0)) return; // Mesh.cpp:750
    const int4 tv = v.tet[t];
    const d3 x0 = ld3(v.x, tv.x), x1 = ld3(v.x, tv.y), x2 = ld3(v.x, tv.z), x3 = ld3(v.x, tv.w);
    const double e[9] = { x1.x - x0.x, x1.y - x0.y, x1.z - x0.z, x2.x - x0.x, x2.y - x0.y, x2.z - x0.z, x3.x - x0.x,
        x3.y - x0.y, x3.z - x0.z };
    if (det3(e) < 0.0) atomicOr(flag, 1);
}

// ---- tetrahedron inversion step bound (get_feasible_steps.cpp:9-172) ----------------------------
struct cplx {
    double re, im;
};
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re }; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return { a.re + b.re, a.im + b.im }; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return { a.re - b.re, a.im - b.im }; }
__device__ __forceinline__ cplx cscale(cplx a, double s) { return { a.re * s, a.im * s }; }
__device__ __forceinline__ cplx cdiv(cplx a, cplx b)
{
    double d = b.re * b.re + b.im * b.im;
    return { (a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d };
}
__device__ __forceinline__ double cabs_(cplx a) { return hypot(a.re, a.im); }
__device__ __forceinline__ cplx csqrt_(cplx z)
{
    if (z.re == 0.0 && z.im == 0.0) return { 0.0, 0.0 };
    double m = cabs_(z);
    double t = sqrt(0.5 * (m + fabs(z.re)));
    if (z.re >= 0.0) return { t, z.im / (2.0 * t) };
    return { fabs(z.im) / (2.0 * t), copysign(t, z.im) };
}
__device__ __forceinline__ cplx cpow_third(cplx z)
{
    if (z.re == 0.0 && z.im == 0.0) return { 0.0, 0.0 };
    if (z.im == 0.0 && z.re > 0.0) return { cbrt(z.re), 0.0 };
    double r = cbrt(cabs_(z)), th = atan2(z.im, z.re) / 3.0;
    return { r * cos(th), r * sin(th) };
}
__device__ double quad_root(double a, double b, double c, double tol)
{
    double t;
    if (fabs(a) <= tol) t = -c / b;
    else {
        double desc = b * b - 4 * a * c;
        if (desc > 0) {
            t = (-b - sqrt(desc)) / (2 * a);
            if (t < 0) t = (-b + sqrt(desc)) / (2 * a);
        }
        else t = -1;
    }
    return t;
}
__device__ double cubic_root(double a, double b, double c, double d, double tol)
{
    double t = -1;
    if (fabs(a) <= tol) return quad_root(b, c, d, tol);
    cplx delta0 = { b * b - 3 * a * c, 0 };
    cplx delta1 = { 2 * b * b * b - 9 * a * b * c + 27 * a * a * d, 0 };
    cplx disc = csqrt_(csub(cmul(delta1, delta1), cscale(cmul(cmul(delta0, delta0), delta0), 4.0)));
    cplx C = cpow_third(cscale(cadd(delta1, disc), 0.5));
    if (cabs_(C) == 0.0) C = cpow_third(cscale(csub(delta1, disc), 0.5));
    const double h = sqrt(3.0) / 2.0;
    cplx u2 = { -0.5, h }, u3 = { -0.5, -h };
    cplx bb = { b, 0 };
    double m3a = -3.0 * a;
    cplx t1 = cscale(cadd(cadd(bb, C), cdiv(delta0, C)), 1.0 / m3a);
    cplx u2C = cmul(u2, C), u3C = cmul(u3, C);
    cplx t2 = cscale(cadd(cadd(bb, u2C), cdiv(delta0, u2C)), 1.0 / m3a);
    cplx t3 = cscale(cadd(cadd(bb, u3C), cdiv(delta0, u3C)), 1.0 / m3a);
    if ((fabs(t1.im) < tol) && (t1.re > 0)) t = t1.re;
    if ((fabs(t2.im) < tol) && (t2.re > 0) && ((t2.re < t) || (t < 0))) t = t2.re;
    if ((fabs(t3.im) < tol) && (t3.re > 0) && ((t3.re < t) || (t < 0))) t = t3.re;
    return t;
}

// tMax: the caller only acts on a root below it (Energy.cpp:565-581: "0 < min < stepSize").  With a t^3 + b t^2 + c t + d >= d + min(a, 0) tMax^3 + min(b, 0) tMax^2 +
// min(c, 0) tMax on [0, tMax], an element whose bound stays above 1 % of d has no root there (nor a near-double one that the closed form below would report as real)
// and skips the closed form -- complex square and cube roots in fp64, 26.8 of the iteration's 2 360 us at 133 K tets when every lane went through it (round 5:
// in a Newton step of the bench no element comes near inversion, whole waves skip).  The minimum over the elements is unchanged whenever it is below tMax.
__global__ __launch_bounds__(BLOCK) void k_inversion_step(ElemView v, const double* __restrict__ p, double slackness, double tMax,
    unsigned long long* __restrict__ outMin)
{
    const int t = v.tetBegin + blockIdx.x * BLOCK + threadIdx.x;
    double res = 1e20;
    if (t < v.tetEnd) {
        const int4 tv = v.tet[t];
        const d3 x0 = ld3(v.x, tv.x), x1 = ld3(v.x, tv.y), x2 = ld3(v.x, tv.z), x3 = ld3(v.x, tv.w);
        const d3 p0 = ld3(p, tv.x), p1 = ld3(p, tv.y), p2 = ld3(p, tv.z), p3 = ld3(p, tv.w);
        const double v0[3] = { x1.x - x0.x, x1.y - x0.y, x1.z - x0.z }, v1[3] = { x2.x - x0.x, x2.y - x0.y, x2.z - x0.z },
                     v2[3] = { x3.x - x0.x, x3.y - x0.y, x3.z - x0.z };
        const double q0[3] = { p1.x - p0.x, p1.y - p0.y, p1.z - p0.z }, q1[3] = { p2.x - p0.x, p2.y - p0.y, p2.z - p0.z },
                     q2[3] = { p3.x - p0.x, p3.y - p0.y, p3.z - p0.z };
        auto cross = [](const double* x, const double* y, double* z) {
            z[0] = x[1] * y[2] - x[2] * y[1];
            z[1] = x[2] * y[0] - x[0] * y[2];
            z[2] = x[0] * y[1] - x[1] * y[0];
        };
        auto dot = [](const double* x, const double* y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; };
        double pxp[3], vxp[3], pxv[3], vxv[3], mix[3];
        cross(q0, q1, pxp);
        cross(v0, q1, vxp);
        cross(q0, v1, pxv);
        cross(v0, v1, vxv);
        for (int c = 0; c < 3; ++c) mix[c] = vxp[c] + pxv[c];
        const double a = dot(q2, pxp);
        const double b = dot(v2, pxp) + dot(q2, mix);
        const double c = dot(q2, vxv) + dot(v2, mix);
        const double d = (1.0 - slackness) * dot(v2, vxv);
        const double lower = d + tMax * (fmin(c, 0.0) + tMax * (fmin(b, 0.0) + tMax * fmin(a, 0.0)));
        if (!(lower > 0.01 * d)) {
            const double r = cubic_root(a, b, c, d, 1.0e-6);
            res = (r >= 0) ? r : 1e20;
        }
    }
    // all results are >= 0, so their bit patterns order like unsigned integers
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) res = fmin(res, __shfl_down(res, off, 64));
    // the caller starts the minimum at 1e20 = "no element limits the step": waves without a root (almost all of them) have nothing to report,
    // and the ones that do are few -- same-address atomics retire one at a time (~2.5 ns each; one per wave was 5 us of this kernel)
    if ((threadIdx.x & 63) == 0 && res < 1e20) atomicMin(outMin, (unsigned long long)__double_as_longlong(res));
}

// ---- nodal helpers -----------------------------------------------------------------------------
__global__ void k_step_forward(int n, const double* __restrict__ x0, const double* __restrict__ p, double alpha, double* __restrict__ x)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = x0[i] + alpha * p[i];
}
// The step size of the first trial of a contact-free line search, decided where the inversion step filter left its result (Energy.cpp:565-581:
// the filter applies when 0 < t < step; Optimizer.cpp:1887), and the trial step taken with it -- without the host in between
__global__ void k_trial_alpha(const double* __restrict__ filterMin, int useFilter, double* __restrict__ alphaOut)
{
    double alpha = 1.0;
    const double t = filterMin[0];
    if (useFilter && t > 0.0 && t < alpha) alpha = t;
    alphaOut[0] = alpha;
}
__global__ void k_step_forward_dev(int n, const double* __restrict__ x0, const double* __restrict__ p, const double* __restrict__ alphaPtr,
    double* __restrict__ x)
{
    const double alpha = alphaPtr[0];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = x0[i] + alpha * p[i];
}
// The three small launches in front of the trial -- copy x -> x0, decide alpha, take the step -- as one (round 4: a dependent launch costs 3.4 us whatever it does):
// every lane applies k_trial_alpha's rule to the same scalar, lane 0 publishes the result
__global__ void k_trial_step_fused(int n, double* __restrict__ x, double* __restrict__ x0, const double* __restrict__ p, const double* __restrict__ filterMin,
    int useFilter, double* __restrict__ alphaOut)
{
    double alpha = 1.0;
    const double t = filterMin[0];
    if (useFilter && t > 0.0 && t < alpha) alpha = t;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) alphaOut[0] = alpha;
    if (i < n) {
        const double xi = x[i];
        x0[i] = xi;
        x[i] = xi + alpha * p[i];
    }
}
// the scalars a contact-free Newton iteration accumulates into behind the solve, reset by one launch: |p|_inf = 0, the step filter's minimum = 1e20, inversion flag = 0
__global__ void k_iter_reset(double* __restrict__ scalar, int* __restrict__ flag)
{
    scalar[3] = 0.0;
    scalar[2] = 1e20;
    flag[0] = 0;
}
// the inversion flag and the scalars to their mapped host buffers in one launch
__global__ void k_publish2(const unsigned* __restrict__ a, unsigned* __restrict__ da, int na, const unsigned* __restrict__ b, unsigned* __restrict__ db, int nb)
{
    const int i = threadIdx.x;
    if (i < na) da[i] = a[i];
    if (i < nb) db[i] = b[i];
}
// grid-stride, one atomic per workgroup (the launch caps the grid at 256 workgroups): per wave of a thread-per-entry launch they were 2 100
// same-address atomics for 1.35e5 entries, most of the kernel's 23 us
__global__ __launch_bounds__(BLOCK) void k_max_abs(int n, const double* __restrict__ v, unsigned long long* __restrict__ out)
{
    __shared__ double sm[BLOCK / 64];
    double m = 0.0;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) m = fmax(m, fabs(v[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < BLOCK / 64; ++w) m = fmax(m, sm[w]);
        atomicMax(out, (unsigned long long)__double_as_longlong(m));
    }
}
__global__ void k_fill(double* p, size_t n, double v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void k_negate(int n, const double* __restrict__ in, double* __restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = -in[i];
}
__global__ void k_colmajor_to_aos(int nV, const double* __restrict__ src, double* __restrict__ dst)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * nV) {
        int v = i / 3, c = i % 3;
        dst[i] = src[v + (size_t)nV * c];
    }
}
__global__ void k_aos_to_colmajor(int nV, const double* __restrict__ src, double* __restrict__ dst)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * nV) {
        int c = i / nV, v = i % nV;
        dst[i] = src[3 * (size_t)v + c];
    }
}
__global__ void k_csr_symv_zero(int n, double* y)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = 0.0;
}
__global__ void k_csr_symv(int nRows, const int* __restrict__ ia, const int* __restrict__ ja, const double* __restrict__ a,
    const double* __restrict__ x, double* __restrict__ y)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nRows) return;
    double acc = 0.0;
    const double xr = x[r];
    for (int k = ia[r]; k < ia[r + 1]; ++k) {
        const int c = ja[k];
        acc += a[k] * x[c];
        if (c != r) atomicAdd(&y[c], a[k] * xr);
    }
    atomicAdd(&y[r], acc);
}
__global__ void k_precond_diag(int nRows, const int* __restrict__ ia, const double* __restrict__ a, const double* __restrict__ in,
    double* __restrict__ out)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nRows) out[r] = in[r] / a[ia[r]]; // the diagonal is the first entry of every upper-CSR row
}
// Newmark update (Optimizer.cpp:582-590, 1259-1277): vel += dt (1 - gamma) acc; acc = (x - xTilde) / (dt^2 beta) + g;
// vel += dt gamma acc; xPrev = x; xTilde = xPrev + dt vel + beta dt^2 g + (1/2 - beta) dt^2 acc (DBC: xPrev)
__global__ void k_nm_update(int nV, const int* __restrict__ dbc, const double* __restrict__ x, double* __restrict__ xPrev,
    double* __restrict__ vel, double* __restrict__ acc, double* __restrict__ dxElastic, double* __restrict__ xTilde, double dt, double beta,
    double gamma, double gx, double gy, double gz)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * nV) return;
    const int v = i / 3, c = i - 3 * v;
    const double g = c == 0 ? gx : (c == 1 ? gy : gz);
    const double dtSq = dt * dt;
    double vl = vel[i] + dt * (1 - gamma) * acc[i];
    const double dx = x[i] - xTilde[i];
    dxElastic[i] = dx; // Optimizer.cpp:583
    double a = dx / (dtSq * beta);
    a += g;
    vl += dt * gamma * a;
    vel[i] = vl;
    acc[i] = a;
    const double xp = x[i];
    xPrev[i] = xp;
    xTilde[i] = dbc[v] != 0 ? xp : xp + (vl * dt + beta * (dtSq * g) + (0.5 - beta) * (dtSq * a));
}
__global__ void k_be_update(int nV, const int* __restrict__ dbc, const double* __restrict__ x, double* __restrict__ xPrev,
    double* __restrict__ vel, double* __restrict__ acc, double* __restrict__ dxElastic, double* __restrict__ xTilde, double dt, double gx,
    double gy, double gz)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * nV) return;
    const int v = i / 3, c = i % 3;
    const double g = (c == 0) ? gx : (c == 1 ? gy : gz);
    const double xi = x[i];
    const double vi = (xi - xPrev[i]) / dt;
    dxElastic[i] = xi - xTilde[i]; // Optimizer.cpp:574
    acc[i] = (vi - vel[i]) / dt; // :577
    vel[i] = vi;
    xPrev[i] = xi;
    xTilde[i] = (dbc[v] != 0) ? xi : (xi + (vi * dt + dt * dt * g));
}
__global__ void k_twist_dir(int nH, const int* __restrict__ ids, const double* __restrict__ ang, double cy, double cz,
    const double* __restrict__ x, double* __restrict__ p)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nH) return;
    const int v = ids[i];
    const double cs = cos(ang[i]), sn = sin(ang[i]);
    const double y = x[3 * (size_t)v + 1] - cy, z = x[3 * (size_t)v + 2] - cz;
    p[3 * (size_t)v + 0] = 0.0;
    p[3 * (size_t)v + 1] = (cs * y - sn * z + cy) - x[3 * (size_t)v + 1];
    p[3 * (size_t)v + 2] = (sn * y + cs * z + cz) - x[3 * (size_t)v + 2];
}

// first iterate of a time step, initX options 1-4 (Optimizer.cpp:936-1080): p = dt v + cg dt^2 g + ce dx_Elastic on the free nodes
__global__ void k_warm_dir(int nV, const int* __restrict__ dbc, const double* __restrict__ vel, const double* __restrict__ dx, double dt,
    double gx, double gy, double gz, double ce, double* __restrict__ p)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * nV) return;
    const int v = i / 3, c = i - 3 * v;
    const double g = c == 0 ? gx : (c == 1 ? gy : gz);
    p[i] = (dbc[v] != 0) ? 0.0 : dt * vel[i] + g + ce * dx[i];
}

// positions of a vertex list, packed (for the bounding box of a Dirichlet group)
__global__ void k_gather3(int n, const int* __restrict__ ids, const double* __restrict__ x, double* __restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n) return;
    const int k = i / 3, c = i - 3 * k;
    out[i] = x[3 * (size_t)ids[k] + c];
}
// target positions of the scripted nodes (AnimScripter.cpp:2150-2157): pos = x + p at the listed nodes, and p at those nodes for the host's tolerance sum
__global__ void k_target_positions(int n, const int* __restrict__ ids, const double* __restrict__ x, const double* __restrict__ p, double* __restrict__ pos,
    double* __restrict__ pOut)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n) return;
    const int k = i / 3, c = i - 3 * k;
    const size_t j = 3 * (size_t)ids[k] + c;
    const double pi = p[j];
    pOut[i] = pi;
    pos[i] = x[j] + pi; // (the same sum the host formed: x[i] += p[i])
}
// scripted Dirichlet motion (AnimScripter.cpp:1440-1462): p += R (x - c) + c + linVel dt - x
__global__ void k_dbc_motion(int n, const int* __restrict__ ids, DbcMotion m, const double* __restrict__ x, double* __restrict__ p)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t v = (size_t)ids[i];
    const double d0 = x[3 * v] - m.c[0], d1 = x[3 * v + 1] - m.c[1], d2 = x[3 * v + 2] - m.c[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) p[3 * v + c] += (m.R[3 * c] * d0 + m.R[3 * c + 1] * d1 + m.R[3 * c + 2] * d2) + m.c[c] + m.linDt[c] - x[3 * v + c];
}

// mesh-sequence Dirichlet motion (AnimScripter.cpp:1465-1528): p = target - x (set, not added: it overrides whatever the velocities asked for)
__global__ void k_dbc_targets(int n, const int* __restrict__ ids, const double* __restrict__ target, const double* __restrict__ x, double* __restrict__ p)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n) return;
    const size_t v = (size_t)ids[i / 3];
    p[3 * v + i % 3] = target[i] - x[3 * v + i % 3];
}

// ---- Neumann boundary conditions (Optimizer.cpp:3241-3250, 3452-3461): nodes `ids`, acceleration a, coefficient dt^2
// gradient: g_v -= dt^2 m_v a ; energy (one workgroup, fixed order): sum dt^2 m_v a . x_v   (non-Dirichlet nodes only)
__global__ void k_nbc_gradient(int n, const int* __restrict__ ids, const int* __restrict__ dbc, const double* __restrict__ mass, double cx, double cy,
    double cz, double* __restrict__ g)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t v = (size_t)ids[i];
    if (dbc[v] != 0) return;
    g[3 * v] -= mass[v] * cx;
    g[3 * v + 1] -= mass[v] * cy;
    g[3 * v + 2] -= mass[v] * cz;
}
__global__ __launch_bounds__(BLOCK) void k_nbc_energy(int n, const int* __restrict__ ids, const int* __restrict__ dbc, const double* __restrict__ mass,
    const double* __restrict__ x, double cx, double cy, double cz, double* __restrict__ out)
{
    __shared__ double sm[BLOCK / 64];
    double acc = 0.0;
    for (int t = threadIdx.x; t < n; t += BLOCK) {
        const size_t v = (size_t)ids[t];
        if (dbc[v] == 0) acc += mass[v] * (x[3 * v] * cx + x[3 * v + 1] * cy + x[3 * v + 2] * cz);
    }
    const double r = block_sum(acc, sm);
    if (threadIdx.x == 0) out[0] = r;
}

// ---- augmented-Lagrangian Dirichlet fallback (AnimScripter.cpp:2303-2346): nodes `ids` with target positions `pos` and
// multipliers `lam` (3 per node)
__global__ void k_clear_projected(int nV, const int* __restrict__ dbc, int projectDBC, double* __restrict__ g)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * nV && projected_dbc(dbc[i / 3], projectDBC)) g[i] = 0.0; // Optimizer.cpp:3512-3516
}
// owner-computes sharding: a nodal vector keeps the entries of the nodes this rank is the designated contributor of (the sum over the ranks is then the vector)
__global__ void k_keep_mine3(int nV, const unsigned char* __restrict__ mine, double* __restrict__ g)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * nV && !mine[i / 3]) g[i] = 0.0;
}
// ... and the CSR values the rows of those nodes (ipcgpu_opt_complete_matrix: the rare consumers of the WHOLE matrix on a sharded context)
__global__ void k_keep_mine_rows(int nRows, const unsigned char* __restrict__ mine, const int* __restrict__ ia, double* __restrict__ a)
{
    const int r = blockIdx.x;
    if (r >= nRows || mine[r / 3]) return;
    for (int k = ia[r] + threadIdx.x; k < ia[r + 1]; k += blockDim.x) a[k] = 0.0;
}
// one workgroup, fixed-order reduction: mode 0  sum -sqrt(m) lam . d + rho / 2 m |d|^2 ; mode 1  sum |d|^2   (d = x - target)
__global__ __launch_bounds__(BLOCK) void k_mdbc_reduce(int n, const int* __restrict__ ids, const double* __restrict__ pos,
    const double* __restrict__ lam, const double* __restrict__ mass, const double* __restrict__ x, double rho, int mode, double* __restrict__ out)
{
    __shared__ double sm[BLOCK / 64];
    double acc = 0.0;
    for (int t = threadIdx.x; t < n; t += BLOCK) {
        const size_t v = (size_t)ids[t];
        double dot = 0.0, sq = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double d = x[3 * v + c] - pos[3 * (size_t)t + c];
            dot += lam[3 * (size_t)t + c] * d;
            sq += d * d;
        }
        acc += mode == 0 ? (rho / 2.0 * mass[v] * sq - sqrt(mass[v]) * dot) : sq;
    }
    const double r = block_sum(acc, sm);
    if (threadIdx.x == 0) out[0] = r;
}
// ---- lagged stiffness-proportional damping (Optimizer.cpp:3381-3400, 3519-3540, 3707-3709, 3723-3735) -----------------------
// displacement of the step with the rows of Dirichlet nodes cleared: mode 0 every Dirichlet node (energy), 1 the projected ones
__global__ void k_damp_dx(int nV, const int* __restrict__ dbc, int mode, int projectDBC, const double* __restrict__ x, const double* __restrict__ xPrev,
    double* __restrict__ dx)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * nV) return;
    const int t = dbc[i / 3];
    const bool zero = mode == 0 ? t != 0 : projected_dbc(t, projectDBC);
    dx[i] = zero ? 0.0 : x[i] - xPrev[i];
}
// the patch pass writes identity / mass on the diagonal of the rows it owns; the damping matrix carries neither
__global__ void k_damp_clear_diag(int nV, const int* __restrict__ dbc, const int* __restrict__ ia, double* __restrict__ d)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * nV && projected_dbc(dbc[i / 3], 1)) d[ia[i]] = 0.0; // the diagonal leads every upper-CSR row
}
__global__ void k_axpy(long long n, double alpha, const double* __restrict__ x, double* __restrict__ y)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += alpha * x[i];
}
// one workgroup, fixed-order reduction: out = scale * x . y
__global__ __launch_bounds__(BLOCK) void k_dot_scaled(int n, const double* __restrict__ x, const double* __restrict__ y, double scale, double* __restrict__ out)
{
    __shared__ double sm[BLOCK / 64];
    double acc = 0.0;
    for (int t = threadIdx.x; t < n; t += BLOCK) acc += x[t] * y[t];
    const double r = block_sum(acc, sm);
    if (threadIdx.x == 0) out[0] = scale * r;
}
// g . e and g . g in one pass over many workgroups (initKappa, Optimizer.cpp:2236-2313: once per time step over 3 nV entries -- as two single-workgroup dot products
// 2 x 93 us at 40 K nodes, 2 x 1.5 ms at 375 K): block b sums its contiguous chunk in a fixed order, k_dot2_final adds the partial sums in block order
__global__ __launch_bounds__(BLOCK) void k_dot2_partial(int n, const double* __restrict__ g, const double* __restrict__ e, double* __restrict__ partial)
{
    __shared__ double sm[BLOCK / 64];
    const int chunk = (n + gridDim.x - 1) / gridDim.x, i0 = blockIdx.x * chunk, i1 = min(n, i0 + chunk);
    double a1 = 0.0, a2 = 0.0;
    for (int i = i0 + threadIdx.x; i < i1; i += BLOCK) {
        const double gi = g[i];
        a1 += gi * e[i];
        a2 += gi * gi;
    }
    const double r1 = block_sum(a1, sm);
    __syncthreads();
    const double r2 = block_sum(a2, sm);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = r1;
        partial[gridDim.x + blockIdx.x] = r2;
    }
}
__global__ void k_dot2_final(int nb, const double* __restrict__ partial, double* __restrict__ out2)
{
    if (threadIdx.x < 2) {
        double s = 0.0;
        for (int b = 0; b < nb; ++b) s += partial[threadIdx.x * nb + b];
        out2[threadIdx.x] = s;
    }
}
__global__ void k_mdbc_gradient(int n, const int* __restrict__ ids, const double* __restrict__ pos, const double* __restrict__ lam,
    const double* __restrict__ mass, const double* __restrict__ x, double rho, double* __restrict__ g)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n) return;
    const int t = i / 3, c = i - 3 * t;
    const size_t v = (size_t)ids[t];
    double gi = g[3 * v + c];
    gi -= sqrt(mass[v]) * lam[i];
    gi += rho * mass[v] * (x[3 * v + c] - pos[i]);
    g[3 * v + c] = gi;
}
__global__ void k_mdbc_hessian(int n, const int* __restrict__ ids, const double* __restrict__ mass, const int* __restrict__ ia, double rho,
    double* __restrict__ a)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n) return;
    const int t = i / 3, c = i - 3 * t;
    const int v = ids[t];
    a[ia[3 * v + c]] += rho * mass[v]; // the diagonal leads every upper-CSR row
}
__global__ void k_mdbc_lambda(int n, const int* __restrict__ ids, const double* __restrict__ pos, double* __restrict__ lam,
    const double* __restrict__ mass, const double* __restrict__ x, double rho)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n) return;
    const int t = i / 3, c = i - 3 * t;
    const size_t v = (size_t)ids[t];
    lam[i] -= rho * sqrt(mass[v]) * (x[3 * v + c] - pos[i]);
}

inline int nblk(long long n, int b = BLOCK) { return (int)((n + b - 1) / b); }

} // namespace

void launch_node_init(const ElemView& v, int projectDBC, bool ownerRank, double* a, double* grad, hipStream_t s)
{
    if (v.nV) hipLaunchKernelGGL(k_node_init, dim3(nblk(v.nV)), dim3(BLOCK), 0, s, v, projectDBC, ownerRank ? 1 : 0, a, grad);
}

void launch_assemble(const ElemView& v, double coef, int projectDBC, double* grad, double* a, hipStream_t s)
{
    const int n = v.tetEnd - v.tetBegin;
    if (n <= 0) return;
    if (a) hipLaunchKernelGGL(k_assemble<true>, dim3(nblk(n)), dim3(BLOCK), 0, s, v, coef, projectDBC, grad, a);
    else hipLaunchKernelGGL(k_assemble<false>, dim3(nblk(n)), dim3(BLOCK), 0, s, v, coef, projectDBC, grad, a);
}
const char* assemble_kernel_name() { return "k_assemble"; }

void launch_energy(const ElemView& v, double coef, bool withInertia, bool ownerRank, double* partial, int partialCap, double* out,
    hipStream_t s)
{
    const int n = std::max(v.tetEnd - v.tetBegin, withInertia ? v.nV : 0);
    const int nb = std::max(1, nblk(n));
    (void)partialCap;
    hipLaunchKernelGGL(k_energy, dim3(nb), dim3(BLOCK), 0, s, v, coef, withInertia ? 1 : 0, ownerRank ? 1 : 0, partial);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(BLOCK), 0, s, partial, nb, out);
}
void launch_energy_per_elem(const ElemView& v, double* perElem, hipStream_t s)
{
    if (v.nT) hipLaunchKernelGGL(k_energy_per_elem, dim3(nblk(v.nT)), dim3(BLOCK), 0, s, v, perElem);
}
void launch_check_inversion(const ElemView& v, int* flag, hipStream_t s)
{
    const int n = v.tetEnd - v.tetBegin;
    if (n > 0) hipLaunchKernelGGL(k_check_inversion, dim3(nblk(n)), dim3(BLOCK), 0, s, v, flag);
}
void launch_inversion_step(const ElemView& v, const double* p, double slackness, double tMax, double* outMin, hipStream_t s)
{
    const int n = v.tetEnd - v.tetBegin;
    if (n > 0)
        hipLaunchKernelGGL(k_inversion_step, dim3(nblk(n)), dim3(BLOCK), 0, s, v, p, slackness, tMax, (unsigned long long*)outMin);
}
void launch_step_forward(int n3, const double* x0, const double* p, double alpha, double* x, hipStream_t s)
{
    if (n3) hipLaunchKernelGGL(k_step_forward, dim3(nblk(n3)), dim3(BLOCK), 0, s, n3, x0, p, alpha, x);
}
void launch_trial_step(int n3, const double* x0, const double* p, const double* filterMin, bool useFilter, double* alphaOut, double* x, hipStream_t s)
{
    hipLaunchKernelGGL(k_trial_alpha, dim3(1), dim3(1), 0, s, filterMin, useFilter ? 1 : 0, alphaOut);
    if (n3) hipLaunchKernelGGL(k_step_forward_dev, dim3(nblk(n3)), dim3(BLOCK), 0, s, n3, x0, p, (const double*)alphaOut, x);
}
void launch_trial_step_fused(int n3, double* x, double* x0, const double* p, const double* filterMin, bool useFilter, double* alphaOut, hipStream_t s)
{
    hipLaunchKernelGGL(k_trial_step_fused, dim3(std::max(nblk(n3), 1)), dim3(BLOCK), 0, s, n3, x, x0, p, filterMin, useFilter ? 1 : 0, alphaOut);
}
void launch_iter_reset(double* scalar, int* flag, hipStream_t s) { hipLaunchKernelGGL(k_iter_reset, dim3(1), dim3(1), 0, s, scalar, flag); }
void launch_publish2(const void* a, void* da, int na, const void* b, void* db, int nb, hipStream_t s)
{
    hipLaunchKernelGGL(k_publish2, dim3(1), dim3(64), 0, s, (const unsigned*)a, (unsigned*)da, na, (const unsigned*)b, (unsigned*)db, nb);
}
void launch_max_abs(int n, const double* v, double* out, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_max_abs, dim3(std::min(nblk(n), 256)), dim3(BLOCK), 0, s, n, v, (unsigned long long*)out);
}
void launch_fill(double* p, size_t n, double v, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_fill, dim3(nblk((long long)n)), dim3(BLOCK), 0, s, p, n, v);
}
void launch_negate(int n, const double* in, double* out, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_negate, dim3(nblk(n)), dim3(BLOCK), 0, s, n, in, out);
}
void launch_colmajor_to_aos(int nV, const double* src, double* dst, hipStream_t s)
{
    if (nV) hipLaunchKernelGGL(k_colmajor_to_aos, dim3(nblk(3LL * nV)), dim3(BLOCK), 0, s, nV, src, dst);
}
void launch_aos_to_colmajor(int nV, const double* src, double* dst, hipStream_t s)
{
    if (nV) hipLaunchKernelGGL(k_aos_to_colmajor, dim3(nblk(3LL * nV)), dim3(BLOCK), 0, s, nV, src, dst);
}
void launch_csr_symv(int nRows, const int* ia, const int* ja, const double* a, const double* x, double* y, hipStream_t s)
{
    if (!nRows) return;
    hipLaunchKernelGGL(k_csr_symv_zero, dim3(nblk(nRows)), dim3(BLOCK), 0, s, nRows, y);
    hipLaunchKernelGGL(k_csr_symv, dim3(nblk(nRows)), dim3(BLOCK), 0, s, nRows, ia, ja, a, x, y);
}
__global__ void k_apply_host_updates(long long nnz, const double* __restrict__ delta, const unsigned char* __restrict__ mask,
    const double* __restrict__ setVal, double* __restrict__ a)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) a[k] = (mask && mask[k]) ? setVal[k] + delta[k] : a[k] + delta[k];
}
void launch_apply_host_updates(long long nnz, const double* delta, const unsigned char* mask, const double* setVal, double* a, hipStream_t s)
{
    if (nnz) hipLaunchKernelGGL(k_apply_host_updates, dim3(nblk(nnz)), dim3(BLOCK), 0, s, nnz, delta, mask, setVal, a);
}
void launch_precond_diag(int nRows, const int* ia, const double* a, const double* in, double* out, hipStream_t s)
{
    if (nRows) hipLaunchKernelGGL(k_precond_diag, dim3(nblk(nRows)), dim3(BLOCK), 0, s, nRows, ia, a, in, out);
}
void launch_nm_update(int nV, const int* dbc, const double* x, double* xPrev, double* vel, double* acc, double* dxElastic, double* xTilde,
    double dt, double beta, double gamma, double gx, double gy, double gz, hipStream_t s)
{
    hipLaunchKernelGGL(k_nm_update, dim3((3 * nV + 255) / 256), dim3(256), 0, s, nV, dbc, x, xPrev, vel, acc, dxElastic, xTilde, dt, beta, gamma,
        gx, gy, gz);
}
void launch_be_update(int nV, const int* dbc, const double* x, double* xPrev, double* vel, double* acc, double* dxElastic, double* xTilde,
    double dt, double gx, double gy, double gz, hipStream_t s)
{
    if (nV)
        hipLaunchKernelGGL(k_be_update, dim3(nblk(3LL * nV)), dim3(BLOCK), 0, s, nV, dbc, x, xPrev, vel, acc, dxElastic, xTilde, dt, gx, gy,
            gz);
}
void launch_warm_dir(int nV, const int* dbc, const double* vel, const double* dx, double dt, const double* cgDtSqG3, double ce, double* p,
    hipStream_t s)
{
    if (nV) hipLaunchKernelGGL(k_warm_dir, dim3(nblk(3LL * nV)), dim3(BLOCK), 0, s, nV, dbc, vel, dx, dt, cgDtSqG3[0], cgDtSqG3[1], cgDtSqG3[2], ce, p);
}
void launch_twist_dir(int nH, const int* ids, const double* ang, double cy, double cz, const double* x, double* p, hipStream_t s)
{
    if (nH) hipLaunchKernelGGL(k_twist_dir, dim3(nblk(nH)), dim3(BLOCK), 0, s, nH, ids, ang, cy, cz, x, p);
}

void launch_nbc_gradient(int n, const int* ids, const int* dbc, const double* mass, const double* dtSqA3, double* g, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_nbc_gradient, dim3(nblk(n)), dim3(BLOCK), 0, s, n, ids, dbc, mass, dtSqA3[0], dtSqA3[1], dtSqA3[2], g);
}
void launch_nbc_energy(int n, const int* ids, const int* dbc, const double* mass, const double* x, const double* dtSqA3, double* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_nbc_energy, dim3(1), dim3(BLOCK), 0, s, n, ids, dbc, mass, x, dtSqA3[0], dtSqA3[1], dtSqA3[2], out);
}
void launch_damp_dx(int nV, const int* dbc, int mode, int projectDBC, const double* x, const double* xPrev, double* dx, hipStream_t s)
{
    if (nV) hipLaunchKernelGGL(k_damp_dx, dim3(nblk(3LL * nV)), dim3(BLOCK), 0, s, nV, dbc, mode, projectDBC, x, xPrev, dx);
}
void launch_damp_clear_diag(int nV, const int* dbc, const int* ia, double* d, hipStream_t s)
{
    if (nV) hipLaunchKernelGGL(k_damp_clear_diag, dim3(nblk(3LL * nV)), dim3(BLOCK), 0, s, nV, dbc, ia, d);
}
void launch_axpy(long long n, double alpha, const double* x, double* y, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_axpy, dim3(nblk(n)), dim3(BLOCK), 0, s, n, alpha, x, y);
}
void launch_dot_scaled(int n, const double* x, const double* y, double scale, double* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_dot_scaled, dim3(1), dim3(BLOCK), 0, s, n, x, y, scale, out);
}
void launch_dot2(int n, const double* g, const double* e, double* partial, int partialCap, double* out2, hipStream_t s)
{
    const int nb = std::max(1, std::min(std::min(128, partialCap / 2), (n + BLOCK - 1) / BLOCK));
    hipLaunchKernelGGL(k_dot2_partial, dim3(nb), dim3(BLOCK), 0, s, n, g, e, partial);
    hipLaunchKernelGGL(k_dot2_final, dim3(1), dim3(64), 0, s, nb, partial, out2);
}
void launch_keep_mine3(int nV, const unsigned char* mine, double* g, hipStream_t s)
{
    if (nV) hipLaunchKernelGGL(k_keep_mine3, dim3(nblk(3LL * nV)), dim3(BLOCK), 0, s, nV, mine, g);
}
void launch_keep_mine_rows(int nRows, const unsigned char* mine, const int* ia, double* a, hipStream_t s)
{
    if (nRows) hipLaunchKernelGGL(k_keep_mine_rows, dim3(nRows), dim3(64), 0, s, nRows, mine, ia, a);
}
void launch_clear_projected(int nV, const int* dbc, int projectDBC, double* g, hipStream_t s)
{
    if (nV) hipLaunchKernelGGL(k_clear_projected, dim3(nblk(3LL * nV)), dim3(BLOCK), 0, s, nV, dbc, projectDBC, g);
}
void launch_mdbc_reduce(const MdbcView& m, const double* x, double rho, int mode, double* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_mdbc_reduce, dim3(1), dim3(BLOCK), 0, s, m.n, m.ids, m.pos, m.lam, m.mass, x, rho, mode, out);
}
void launch_mdbc_gradient(const MdbcView& m, const double* x, double rho, double* g, hipStream_t s)
{
    if (m.n) hipLaunchKernelGGL(k_mdbc_gradient, dim3(nblk(3LL * m.n)), dim3(BLOCK), 0, s, m.n, m.ids, m.pos, m.lam, m.mass, x, rho, g);
}
void launch_mdbc_hessian(const MdbcView& m, const int* ia, double rho, double* a, hipStream_t s)
{
    if (m.n) hipLaunchKernelGGL(k_mdbc_hessian, dim3(nblk(3LL * m.n)), dim3(BLOCK), 0, s, m.n, m.ids, m.mass, ia, rho, a);
}
void launch_mdbc_lambda(const MdbcView& m, const double* x, double rho, hipStream_t s)
{
    if (m.n) hipLaunchKernelGGL(k_mdbc_lambda, dim3(nblk(3LL * m.n)), dim3(BLOCK), 0, s, m.n, m.ids, m.pos, m.lam, m.mass, x, rho);
}
void launch_target_positions(int n, const int* ids, const double* x, const double* p, double* pos, double* pOut, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_target_positions, dim3(nblk(3LL * n)), dim3(BLOCK), 0, s, n, ids, x, p, pos, pOut);
}
void launch_gather3(int n, const int* ids, const double* x, double* out, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_gather3, dim3(nblk(3LL * n)), dim3(BLOCK), 0, s, n, ids, x, out);
}
void launch_dbc_motion(int n, const int* ids, const DbcMotion& m, const double* x, double* p, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_dbc_motion, dim3(nblk(n)), dim3(BLOCK), 0, s, n, ids, m, x, p);
}
void launch_dbc_targets(int n, const int* ids, const double* target, const double* x, double* p, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_dbc_targets, dim3(nblk(3LL * n)), dim3(BLOCK), 0, s, n, ids, target, x, p);
}

__global__ void k_publish(const unsigned* __restrict__ src, unsigned* __restrict__ dst, int n)
{
    const int i = threadIdx.x;
    if (i < n) dst[i] = src[i];
}
void launch_publish(const void* src_dev, void* dst_mapped, int nWords, hipStream_t s)
{
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s, (const unsigned*)src_dev, (unsigned*)dst_mapped, nWords);
}

} // namespace ipcgpu
