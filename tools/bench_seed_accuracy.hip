// accuracy of the v_rcp_f64 / v_rsq_f64 seeds on gfx950 and after one / two Newton steps (what the Jacobi rotations and the pivot chain may rely on)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* o, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    double r = __builtin_amdgcn_rcp(d);
    o[6 * i + 0] = r;
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    o[6 * i + 1] = r;
    e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    o[6 * i + 2] = r;
    double q = __builtin_amdgcn_rsq(d);
    o[6 * i + 3] = q;
    double f = fma(-d * q, q, 1.0);
    q = fma(0.5 * q, f, q);
    o[6 * i + 4] = q;
    f = fma(-d * q, q, 1.0);
    q = fma(0.5 * q, f, q);
    o[6 * i + 5] = q;
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> h(n), o(6 * (size_t)n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;
        h[i] = std::exp((u - 0.5) * 40.0); // 1e-9 .. 1e9
    }
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * (size_t)n * 8);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(o.data(), dout, 6 * (size_t)n * 8, hipMemcpyDeviceToHost);
    double m[6] = { 0, 0, 0, 0, 0, 0 };
    for (int i = 0; i < n; ++i) {
        const long double d = h[i];
        for (int j = 0; j < 3; ++j) m[j] = std::fmax(m[j], (double)fabsl((long double)o[6 * i + j] * d - 1.0L));
        for (int j = 3; j < 6; ++j) m[j] = std::fmax(m[j], (double)fabsl((long double)o[6 * i + j] * sqrtl(d) - 1.0L));
    }
    printf("v_rcp_f64 max relative error: seed %.3e, one Newton step %.3e, two %.3e\n", m[0], m[1], m[2]);
    printf("v_rsq_f64 max relative error: seed %.3e, one Newton step %.3e, two %.3e   (2^-53 = %.3e)\n", m[3], m[4], m[5], std::ldexp(1.0, -53));
    return 0;
}
