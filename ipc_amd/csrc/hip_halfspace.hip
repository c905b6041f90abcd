// Half-space obstacle kernels.  Per-vertex streaming work: one lane per surface vertex / per
// constraint, 24 B of position per vertex, so every kernel here is a single coalesced pass bounded by HBM latency at
// the sizes of the hot path (<= 45 K surface vertices at mat150).  Reference lines are cited in hip_halfspace.h.
#include "hip_halfspace.h"
#include "contact_device.h"
#include <cmath>
#include <cstring>
#include <hipcub/hipcub.hpp>

namespace ipcgpu {
namespace {
constexpr int BLOCK = 256;
inline int nblk(int n) { return (n + BLOCK - 1) / BLOCK; }

struct Plane {
    double n0, n1, n2, D;
};
__device__ __forceinline__ double plane_dist(const Plane& h, const double* __restrict__ x, int v)
{
    return h.n0 * x[3 * (size_t)v] + h.n1 * x[3 * (size_t)v + 1] + h.n2 * x[3 * (size_t)v + 2] + h.D;
}
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
// order-preserving map double -> uint64 so that atomicMin works for either sign
__device__ __forceinline__ unsigned long long ordered_bits(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
inline double from_ordered_bits(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    double v;
    std::memcpy(&v, &b, sizeof(v));
    return v;
}

__global__ void k_hs_flags(int nSVI, const int* __restrict__ svi, const double* __restrict__ x, const int* __restrict__ dbc, Plane h, double dHat,
    int* __restrict__ flags)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nSVI) return;
    const int v = svi[i];
    const double dist = plane_dist(h, x, v);
    flags[i] = (dbc[v] == 0 && dist * dist < dHat) ? 1 : 0;
}

__global__ __launch_bounds__(BLOCK) void k_hs_energy(int n, const int* __restrict__ set, const double* __restrict__ x, Plane h, double dHat,
    double* __restrict__ partial)
{
    __shared__ double sm[BLOCK / 64];
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    double val = 0.0;
    if (i < n) {
        const double dist = plane_dist(h, x, set[i]);
        double b, gb, Hb;
        cdev::barrier(dist * dist, dHat, &b, &gb, &Hb);
        val = b;
    }
    const double r = block_sum(val, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
__global__ __launch_bounds__(BLOCK) void k_hs_reduce(const double* __restrict__ partial, int n, double scale, double* __restrict__ out)
{
    __shared__ double sm[BLOCK / 64];
    double x = 0.0;
    for (int i = threadIdx.x; i < n; i += BLOCK) x += partial[i];
    const double r = block_sum(x, sm);
    if (threadIdx.x == 0) out[0] = scale * r;
}

__global__ void k_hs_gradient(int n, const int* __restrict__ set, const double* __restrict__ x, Plane h, double dHat, double kappa,
    double* __restrict__ grad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int v = set[i]; // vertices of one set are distinct: plain read-modify-write
    const double dist = plane_dist(h, x, v);
    double b, gb, Hb;
    cdev::barrier(dist * dist, dHat, &b, &gb, &Hb);
    grad[3 * (size_t)v] += kappa * gb * 2.0 * dist * h.n0;
    grad[3 * (size_t)v + 1] += kappa * gb * 2.0 * dist * h.n1;
    grad[3 * (size_t)v + 2] += kappa * gb * 2.0 * dist * h.n2;
}

__global__ void k_hs_hessian(int n, const int* __restrict__ set, const double* __restrict__ x, const int* __restrict__ dbc,
    const int* __restrict__ rowBase, const int* __restrict__ rowLen, Plane h, double dHat, double kappa, int projectDBC, double* __restrict__ a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int v = set[i];
    if (dbc[v] != 0 && projectDBC) return;
    const double dist = plane_dist(h, x, v), d = dist * dist;
    double b, gb, Hb;
    cdev::barrier(d, dHat, &b, &gb, &Hb);
    const double param = 4.0 * Hb * d + 2.0 * gb;
    if (!(param > 0.0)) return;
    const double nn[3] = { h.n0, h.n1, h.n2 };
    const int p0 = rowBase[v], L = rowLen[v]; // upper triangle of the diagonal 3x3 block (LinSysSolver.hpp:63-111)
    a[p0] += kappa * param * nn[0] * nn[0];
    a[p0 + 1] += kappa * param * nn[0] * nn[1];
    a[p0 + 2] += kappa * param * nn[0] * nn[2];
    a[p0 + L] += kappa * param * nn[1] * nn[1];
    a[p0 + L + 1] += kappa * param * nn[1] * nn[2];
    a[p0 + 2 * L - 1] += kappa * param * nn[2] * nn[2];
}

__global__ void k_hs_step(int nSVI, const int* __restrict__ svi, const double* __restrict__ x, const int* __restrict__ dbc, const double* __restrict__ p,
    Plane h, double slackness, unsigned long long* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double best = 1.0;
    if (i < nSVI) {
        const int v = svi[i];
        if (dbc[v] == 0) {
            const double coef = h.n0 * p[3 * (size_t)v] + h.n1 * p[3 * (size_t)v + 1] + h.n2 * p[3 * (size_t)v + 2];
            if (coef < 0.0) best = -plane_dist(h, x, v) / coef * slackness;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = fmin(best, __shfl_down(best, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMin(out, ordered_bits(best));
}

// the plane moves by delta: every surface node (Dirichlet or not) bounds the fraction, coef = n . (-delta) is the same for all (HalfSpace.cpp:393-411)
__global__ void k_hs_move(int nSVI, const int* __restrict__ svi, const double* __restrict__ x, Plane h, double coef, double slackness, unsigned long long* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double best = 1.0;
    if (i < nSVI) best = -plane_dist(h, x, svi[i]) / coef * slackness;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = fmin(best, __shfl_down(best, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMin(out, ordered_bits(best));
}

__global__ void k_hs_intersected(int nV, const double* __restrict__ x, const int* __restrict__ dbc, Plane h, int* __restrict__ flag)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nV || dbc[v] != 0) return;
    const double dist = plane_dist(h, x, v);
    if (dist * dist <= 0.0) flag[0] = 1;
}

__global__ void k_hs_dist2(int n, const int* __restrict__ verts, const double* __restrict__ x, Plane h, double* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double dist = plane_dist(h, x, verts[i]);
    out[i] = dist * dist;
}

// ---- lagged friction (HalfSpace.cpp:272-381) ------------------------------------------------------------------------
__global__ void k_hs_lag(int n, const int* __restrict__ set, const double* __restrict__ x, Plane h, double dHat, double kappa, double* __restrict__ lambda)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double dist = plane_dist(h, x, set[i]), d = dist * dist;
    double b, gb, Hb;
    cdev::barrier(d, dHat, &b, &gb, &Hb);
    lambda[i] = gb * (-kappa * 2.0 * sqrt(d)); // Optimizer.cpp:1563-1566
}
__device__ __forceinline__ void hs_proj(const Plane& h, const double* __restrict__ x, const double* __restrict__ xt, int v, double* vp)
{
    const double vd[3] = { x[3 * (size_t)v] - xt[3 * (size_t)v], x[3 * (size_t)v + 1] - xt[3 * (size_t)v + 1], x[3 * (size_t)v + 2] - xt[3 * (size_t)v + 2] };
    const double dn = vd[0] * h.n0 + vd[1] * h.n1 + vd[2] * h.n2;
    vp[0] = vd[0] - dn * h.n0;
    vp[1] = vd[1] - dn * h.n1;
    vp[2] = vd[2] - dn * h.n2;
}
__global__ __launch_bounds__(BLOCK) void k_hs_fric_energy(int n, const int* __restrict__ set, const double* __restrict__ lambda, const double* __restrict__ x,
    const double* __restrict__ xt, Plane h, double mu, double eps2, double* __restrict__ partial)
{
    __shared__ double sm[BLOCK / 64];
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    double val = 0.0;
    if (i < n) {
        double vp[3];
        hs_proj(h, x, xt, set[i], vp);
        const double m2 = vp[0] * vp[0] + vp[1] * vp[1] + vp[2] * vp[2], eps = sqrt(eps2);
        val = (m2 > eps2) ? mu * lambda[i] * (sqrt(m2) - eps * 0.5) : mu * lambda[i] * m2 / eps * 0.5;
    }
    const double r = block_sum(val, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
__global__ void k_hs_fric_gradient(int n, const int* __restrict__ set, const double* __restrict__ lambda, const double* __restrict__ x,
    const double* __restrict__ xt, Plane h, double mu, double eps2, double* __restrict__ grad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int v = set[i];
    double vp[3];
    hs_proj(h, x, xt, v, vp);
    const double m2 = vp[0] * vp[0] + vp[1] * vp[1] + vp[2] * vp[2];
    const double sc = (m2 > eps2) ? mu * lambda[i] / sqrt(m2) : mu * lambda[i] / sqrt(eps2);
    for (int c = 0; c < 3; ++c) grad[3 * (size_t)v + c] += sc * vp[c];
}
// sliding: ml/|v| (P - vhat vhat^T) with P = I - n n^T has eigenvalues {ml/|v|, 0, 0}: the reference's makePD (HalfSpace.cpp:356)
// is the identity up to round-off; sticking: ml/eps P ("already SPD", :360)
__global__ void k_hs_fric_hessian(int n, const int* __restrict__ set, const double* __restrict__ lambda, const double* __restrict__ x,
    const double* __restrict__ xt, const int* __restrict__ dbc, const int* __restrict__ rowBase, const int* __restrict__ rowLen, Plane h, double mu,
    double eps2, int projectDBC, double* __restrict__ a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int v = set[i];
    if (projectDBC && dbc[v] != 0) return;
    const double ml = mu * lambda[i];
    double vp[3];
    hs_proj(h, x, xt, v, vp);
    const double m2 = vp[0] * vp[0] + vp[1] * vp[1] + vp[2] * vp[2];
    const double nn[3] = { h.n0, h.n1, h.n2 };
    double H[3][3];
    if (m2 > eps2) {
        const double mag = sqrt(m2);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) H[r][c] = vp[r] * (-ml / m2 / mag) * vp[c] + ((r == c ? 1.0 : 0.0) - nn[r] * nn[c]) * (ml / mag);
    }
    else
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) H[r][c] = ((r == c ? 1.0 : 0.0) - nn[r] * nn[c]) * (ml / sqrt(eps2));
    const int p0 = rowBase[v], L = rowLen[v];
    a[p0] += H[0][0];
    a[p0 + 1] += H[0][1];
    a[p0 + 2] += H[0][2];
    a[p0 + L] += H[1][1];
    a[p0 + L + 1] += H[1][2];
    a[p0 + 2 * L - 1] += H[2][2];
}
} // namespace

HipHalfSpace::HipHalfSpace(hipStream_t s, const double* origin, const double* normal) : stream(s)
{
    const double len = std::sqrt(normal[0] * normal[0] + normal[1] * normal[1] + normal[2] * normal[2]);
    if (!(len > 0.0)) throw ArgError("half-space normal must be non-zero");
    for (int c = 0; c < 3; ++c) n[c] = normal[c] / len;
    for (int c = 0; c < 3; ++c) this->origin[c] = origin[c];
    D = -(n[0] * origin[0] + n[1] * origin[1] + n[2] * origin[2]);
}

int HipHalfSpace::build(int nSVI, const int* svi_dev, const double* x_dev, const int* dbc_dev, double dHat)
{
    set.clear();
    if (!nSVI) return 0;
    const Plane h{ n[0], n[1], n[2], D };
    flags_.alloc(nSVI);
    d_set.alloc(nSVI);
    count_.alloc(1);
    hipLaunchKernelGGL(k_hs_flags, dim3(nblk(nSVI)), dim3(BLOCK), 0, stream, nSVI, svi_dev, x_dev, dbc_dev, h, dHat, flags_.p);
    size_t bytes = 0; // order-preserving compaction: the set comes out in ascending surface-vertex order like the reference's
    HIP_CHECK(hipcub::DeviceSelect::Flagged(nullptr, bytes, svi_dev, flags_.p, d_set.p, count_.p, nSVI, stream));
    if (tmp_.n < bytes) tmp_.alloc(bytes);
    HIP_CHECK(hipcub::DeviceSelect::Flagged(tmp_.p, bytes, svi_dev, flags_.p, d_set.p, count_.p, nSVI, stream));
    int cnt = 0;
    count_.download(&cnt, 1, stream);
    set.resize(cnt);
    if (cnt) d_set.download(set.data(), cnt, stream);
    return cnt;
}

void HipHalfSpace::setSet(int cnt, const int* verts)
{
    set.assign(verts, verts + cnt);
    if (d_set.n < (size_t)cnt) d_set.alloc(cnt);
    if (cnt) HIP_CHECK(hipMemcpyAsync(d_set.p, set.data(), cnt * sizeof(int), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
}

double HipHalfSpace::energy(const double* x_dev, double dHat, double kappa)
{
    const int cnt = (int)set.size();
    if (!cnt) return 0.0;
    const Plane h{ n[0], n[1], n[2], D };
    const int nb = nblk(cnt);
    if (partial_.n < (size_t)nb + 1) partial_.alloc(nb + 1);
    hipLaunchKernelGGL(k_hs_energy, dim3(nb), dim3(BLOCK), 0, stream, cnt, d_set.p, x_dev, h, dHat, partial_.p + 1);
    hipLaunchKernelGGL(k_hs_reduce, dim3(1), dim3(BLOCK), 0, stream, partial_.p + 1, nb, kappa, partial_.p);
    double out = 0.0;
    partial_.download(&out, 1, stream);
    return out;
}

void HipHalfSpace::gradientAdd(const double* x_dev, double dHat, double kappa, double* grad_dev)
{
    const int cnt = (int)set.size();
    if (!cnt) return;
    const Plane h{ n[0], n[1], n[2], D };
    hipLaunchKernelGGL(k_hs_gradient, dim3(nblk(cnt)), dim3(BLOCK), 0, stream, cnt, d_set.p, x_dev, h, dHat, kappa, grad_dev);
}

void HipHalfSpace::hessianAdd(const double* x_dev, const int* dbc_dev, const int* rowBase_dev, const int* rowLen_dev, double dHat, double kappa,
    int projectDBC, double* a_dev)
{
    const int cnt = (int)set.size();
    if (!cnt) return;
    const Plane h{ n[0], n[1], n[2], D };
    hipLaunchKernelGGL(k_hs_hessian, dim3(nblk(cnt)), dim3(BLOCK), 0, stream, cnt, d_set.p, x_dev, dbc_dev, rowBase_dev, rowLen_dev, h, dHat, kappa,
        projectDBC, a_dev);
}

double HipHalfSpace::stepBound(int nSVI, const int* svi_dev, const double* x_dev, const int* dbc_dev, const double* p_dev, double slackness,
    double stepSize)
{
    if (!nSVI) return stepSize;
    const Plane h{ n[0], n[1], n[2], D };
    minOut_.alloc(1);
    const unsigned long long init = ~0ull;
    HIP_CHECK(hipMemcpyAsync(minOut_.p, &init, sizeof(init), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_hs_step, dim3(nblk(nSVI)), dim3(BLOCK), 0, stream, nSVI, svi_dev, x_dev, dbc_dev, p_dev, h, slackness, minOut_.p);
    unsigned long long k = 0;
    minOut_.download(&k, 1, stream);
    return std::min(stepSize, from_ordered_bits(k));
}

double HipHalfSpace::move(int nSVI, const int* svi_dev, const double* x_dev, const double* delta, double slackness)
{
    const double coef = -(n[0] * delta[0] + n[1] * delta[1] + n[2] * delta[2]);
    double stepSize = 1.0;
    if (coef < 0.0 && nSVI) { // going towards the object
        const Plane h{ n[0], n[1], n[2], D };
        minOut_.alloc(1);
        const unsigned long long init = ~0ull;
        HIP_CHECK(hipMemcpyAsync(minOut_.p, &init, sizeof(init), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(k_hs_move, dim3(nblk(nSVI)), dim3(BLOCK), 0, stream, nSVI, svi_dev, x_dev, h, coef, slackness, minOut_.p);
        unsigned long long k = 0;
        minOut_.download(&k, 1, stream);
        stepSize = std::min(1.0, from_ordered_bits(k));
    }
    for (int c = 0; c < 3; ++c) origin[c] += stepSize * delta[c];
    D = -(n[0] * origin[0] + n[1] * origin[1] + n[2] * origin[2]); // init(origin + stepSize * deltaX, normal, ...), HalfSpace.cpp:413
    return 1.0 - stepSize;
}

bool HipHalfSpace::intersected(int nV, const double* x_dev, const int* dbc_dev)
{
    const Plane h{ n[0], n[1], n[2], D };
    count_.alloc(1);
    count_.zero(stream);
    hipLaunchKernelGGL(k_hs_intersected, dim3(nblk(nV)), dim3(BLOCK), 0, stream, nV, x_dev, dbc_dev, h, count_.p);
    int f = 0;
    count_.download(&f, 1, stream);
    return f != 0;
}

void HipHalfSpace::evalDist2(const std::vector<int>& verts, const double* x_dev, std::vector<double>& d2)
{
    const int cnt = (int)verts.size();
    d2.assign(cnt, 0.0);
    if (!cnt) return;
    const Plane h{ n[0], n[1], n[2], D };
    ids_.upload(verts, stream);
    if (vals_.n < (size_t)cnt) vals_.alloc(cnt);
    hipLaunchKernelGGL(k_hs_dist2, dim3(nblk(cnt)), dim3(BLOCK), 0, stream, cnt, ids_.p, x_dev, h, vals_.p);
    vals_.download(d2.data(), cnt, stream);
}

void HipHalfSpace::lagUpdate(const double* x_dev, double dHat, double kappa)
{
    lagSet = set;
    const int cnt = (int)lagSet.size();
    if (!cnt) return;
    const Plane h{ n[0], n[1], n[2], D };
    d_lagSet.ensure(cnt);
    d_lagLambda.ensure(cnt);
    HIP_CHECK(hipMemcpyAsync(d_lagSet.p, d_set.p, cnt * sizeof(int), hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(k_hs_lag, dim3(nblk(cnt)), dim3(BLOCK), 0, stream, cnt, d_lagSet.p, x_dev, h, dHat, kappa, d_lagLambda.p);
}

double HipHalfSpace::frictionEnergy(const double* x_dev, const double* xt_dev, double eps2)
{
    const int cnt = (int)lagSet.size();
    if (!cnt) return 0.0;
    const Plane h{ n[0], n[1], n[2], D };
    const int nb = nblk(cnt);
    if (partial_.n < (size_t)nb + 1) partial_.alloc(nb + 1);
    hipLaunchKernelGGL(k_hs_fric_energy, dim3(nb), dim3(BLOCK), 0, stream, cnt, d_lagSet.p, d_lagLambda.p, x_dev, xt_dev, h, friction, eps2, partial_.p + 1);
    hipLaunchKernelGGL(k_hs_reduce, dim3(1), dim3(BLOCK), 0, stream, partial_.p + 1, nb, 1.0, partial_.p);
    double out = 0.0;
    partial_.download(&out, 1, stream);
    return out;
}

void HipHalfSpace::frictionGradientAdd(const double* x_dev, const double* xt_dev, double eps2, double* grad_dev)
{
    const int cnt = (int)lagSet.size();
    if (!cnt) return;
    const Plane h{ n[0], n[1], n[2], D };
    hipLaunchKernelGGL(k_hs_fric_gradient, dim3(nblk(cnt)), dim3(BLOCK), 0, stream, cnt, d_lagSet.p, d_lagLambda.p, x_dev, xt_dev, h, friction, eps2, grad_dev);
}

void HipHalfSpace::frictionHessianAdd(const double* x_dev, const double* xt_dev, const int* dbc_dev, const int* rowBase_dev, const int* rowLen_dev,
    double eps2, int projectDBC, double* a_dev)
{
    const int cnt = (int)lagSet.size();
    if (!cnt) return;
    const Plane h{ n[0], n[1], n[2], D };
    hipLaunchKernelGGL(k_hs_fric_hessian, dim3(nblk(cnt)), dim3(BLOCK), 0, stream, cnt, d_lagSet.p, d_lagLambda.p, x_dev, xt_dev, dbc_dev, rowBase_dev,
        rowLen_dev, h, friction, eps2, projectDBC, a_dev);
}

} // namespace ipcgpu
