// Exact sign of the orientation determinant of four points -- what a build of the reference with USE_PREDICATES asks of
// igl::predicates::orient3d (Shewchuk's robust predicate; libigl's `predicates` module is a fetched dependency, not part of /root/reference)
// inside IglUtils::segTriIntersect (IglUtils.hpp:222-233) and IglUtils::pointInsideTetrahedron (:280-294).
//
//   orient3d(a, b, c, d) = sign det [a - d; b - d; c - d]      (> 0: d below the plane through a, b, c taken counter-clockwise from above)
//
// Restated from the published algorithm (J. R. Shewchuk, "Adaptive Precision Floating-Point Arithmetic and Fast Robust Geometric Predicates",
// 1997): a floating-point evaluation with the paper's forward error bound as a filter, and, when the filter cannot decide, the determinant
// of the ORIGINAL coordinates as a non-overlapping expansion (error-free product by fused multiply-add, error-free sum, sum / scale of
// expansions with zero elimination): its most significant component carries the sign.  No adaptive stages in between: the exact path is rare
// (nearly coplanar configurations only) and correctness, not its speed, is what the intersection checks need.
//
// Compiles for the device and for the host (tests/test_orient3d.py builds it with g++ and pins it on exact rational arithmetic; the CPU checker of
// the test suite includes it too).
#pragma once
#include <cmath>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define O3_HD __host__ __device__ inline
#else
#define O3_HD inline
#endif

namespace ipcgpu {
namespace o3 {

O3_HD void two_sum(double a, double b, double& x, double& y)
{
    x = a + b;
    const double bv = x - a, av = x - bv;
    y = (a - av) + (b - bv);
}
O3_HD void two_diff(double a, double b, double& x, double& y)
{
    x = a - b;
    const double bv = a - x, av = x + bv;
    y = (a - av) + (bv - b);
}
O3_HD void two_prod(double a, double b, double& x, double& y)
{
    x = a * b;
    y = fma(a, b, -x); // exact: the rounding error of a product is representable
}
// (a1 + a0) - (b1 + b0) as a four-component expansion x[0..3], increasing magnitude
O3_HD void two_two_diff(double a1, double a0, double b1, double b0, double* x)
{
    double i, j, r0, t;
    two_diff(a0, b0, i, x[0]);
    two_sum(a1, i, j, r0);
    two_diff(r0, b1, i, x[1]);
    two_sum(j, i, t, x[2]);
    x[3] = t;
}
// h = e + f for two non-overlapping expansions in increasing magnitude; zero components dropped; returns the length of h
O3_HD int expansion_sum_zeroelim(int elen, const double* e, int flen, const double* f, double* h)
{
    // merge by magnitude, then a running error-free sum (fast-expansion-sum of the paper, figure 4.9, with zero elimination)
    int ei = 0, fi = 0, hlen = 0;
    double Q, q, hh;
    double enow = e[0], fnow = f[0];
    if ((fnow > enow) == (fnow > -enow)) {
        Q = enow;
        ++ei;
    }
    else {
        Q = fnow;
        ++fi;
    }
    if (ei < elen && fi < flen) {
        enow = e[ei];
        fnow = f[fi];
        double nxt;
        if ((fnow > enow) == (fnow > -enow)) {
            nxt = enow;
            ++ei;
        }
        else {
            nxt = fnow;
            ++fi;
        }
        // fast two-sum is not safe here in general: use two_sum throughout (same result, a few more operations)
        two_sum(nxt, Q, q, hh);
        Q = q;
        if (hh != 0.0) h[hlen++] = hh;
        while (ei < elen && fi < flen) {
            enow = e[ei];
            fnow = f[fi];
            if ((fnow > enow) == (fnow > -enow)) {
                nxt = enow;
                ++ei;
            }
            else {
                nxt = fnow;
                ++fi;
            }
            two_sum(Q, nxt, q, hh);
            Q = q;
            if (hh != 0.0) h[hlen++] = hh;
        }
    }
    while (ei < elen) {
        two_sum(Q, e[ei++], q, hh);
        Q = q;
        if (hh != 0.0) h[hlen++] = hh;
    }
    while (fi < flen) {
        two_sum(Q, f[fi++], q, hh);
        Q = q;
        if (hh != 0.0) h[hlen++] = hh;
    }
    if (Q != 0.0 || hlen == 0) h[hlen++] = Q;
    return hlen;
}
// h = b * e (scale-expansion of the paper, figure 4.13, with zero elimination); returns the length of h (<= 2 elen)
O3_HD int scale_expansion_zeroelim(int elen, const double* e, double b, double* h)
{
    double Q, sum, hh, p1, p0;
    int hlen = 0;
    two_prod(e[0], b, Q, hh);
    if (hh != 0.0) h[hlen++] = hh;
    for (int i = 1; i < elen; ++i) {
        two_prod(e[i], b, p1, p0);
        two_sum(Q, p0, sum, hh);
        if (hh != 0.0) h[hlen++] = hh;
        // fast two-sum (p1, sum): |p1| >= |sum| holds for a non-overlapping e in increasing magnitude
        Q = p1 + sum;
        hh = sum - (Q - p1);
        if (hh != 0.0) h[hlen++] = hh;
    }
    if (Q != 0.0 || hlen == 0) h[hlen++] = Q;
    return hlen;
}

// sign (+1 / 0 / -1) of det [a - d; b - d; c - d], exactly
O3_HD int orient3d_exact(const double* pa, const double* pb, const double* pc, const double* pd)
{
    double axby1, axby0, bxay1, bxay0, bxcy1, bxcy0, cxby1, cxby0, cxdy1, cxdy0, dxcy1, dxcy0;
    double dxay1, dxay0, axdy1, axdy0, axcy1, axcy0, cxay1, cxay0, bxdy1, bxdy0, dxby1, dxby0;
    double ab[4], bc[4], cd[4], da[4], ac[4], bd[4];
    two_prod(pa[0], pb[1], axby1, axby0);
    two_prod(pb[0], pa[1], bxay1, bxay0);
    two_two_diff(axby1, axby0, bxay1, bxay0, ab);
    two_prod(pb[0], pc[1], bxcy1, bxcy0);
    two_prod(pc[0], pb[1], cxby1, cxby0);
    two_two_diff(bxcy1, bxcy0, cxby1, cxby0, bc);
    two_prod(pc[0], pd[1], cxdy1, cxdy0);
    two_prod(pd[0], pc[1], dxcy1, dxcy0);
    two_two_diff(cxdy1, cxdy0, dxcy1, dxcy0, cd);
    two_prod(pd[0], pa[1], dxay1, dxay0);
    two_prod(pa[0], pd[1], axdy1, axdy0);
    two_two_diff(dxay1, dxay0, axdy1, axdy0, da);
    two_prod(pa[0], pc[1], axcy1, axcy0);
    two_prod(pc[0], pa[1], cxay1, cxay0);
    two_two_diff(axcy1, axcy0, cxay1, cxay0, ac);
    two_prod(pb[0], pd[1], bxdy1, bxdy0);
    two_prod(pd[0], pb[1], dxby1, dxby0);
    two_two_diff(bxdy1, bxdy0, dxby1, dxby0, bd);
    double t8[8], cda[12], dab[12], abc[12], bcd[12];
    int n8, ncda, ndab, nabc, nbcd;
    n8 = expansion_sum_zeroelim(4, cd, 4, da, t8);
    ncda = expansion_sum_zeroelim(n8, t8, 4, ac, cda);
    n8 = expansion_sum_zeroelim(4, da, 4, ab, t8);
    ndab = expansion_sum_zeroelim(n8, t8, 4, bd, dab);
    for (int i = 0; i < 4; ++i) {
        bd[i] = -bd[i];
        ac[i] = -ac[i];
    }
    n8 = expansion_sum_zeroelim(4, ab, 4, bc, t8);
    nabc = expansion_sum_zeroelim(n8, t8, 4, ac, abc);
    n8 = expansion_sum_zeroelim(4, bc, 4, cd, t8);
    nbcd = expansion_sum_zeroelim(n8, t8, 4, bd, bcd);
    double adet[24], bdet[24], cdet[24], ddet[24], abdet[48], cddet[48], deter[96];
    const int na = scale_expansion_zeroelim(nbcd, bcd, pa[2], adet);
    const int nb = scale_expansion_zeroelim(ncda, cda, -pb[2], bdet);
    const int nc = scale_expansion_zeroelim(ndab, dab, pc[2], cdet);
    const int nd = scale_expansion_zeroelim(nabc, abc, -pd[2], ddet);
    const int nab = expansion_sum_zeroelim(na, adet, nb, bdet, abdet);
    const int ncd = expansion_sum_zeroelim(nc, cdet, nd, ddet, cddet);
    const int n = expansion_sum_zeroelim(nab, abdet, ncd, cddet, deter);
    const double top = deter[n - 1];
    return top > 0.0 ? 1 : (top < 0.0 ? -1 : 0);
}

// the same sign with a floating-point filter in front (the paper's error bound A of orient3d)
O3_HD int orient3d(const double* pa, const double* pb, const double* pc, const double* pd)
{
    const double adx = pa[0] - pd[0], bdx = pb[0] - pd[0], cdx = pc[0] - pd[0];
    const double ady = pa[1] - pd[1], bdy = pb[1] - pd[1], cdy = pc[1] - pd[1];
    const double adz = pa[2] - pd[2], bdz = pb[2] - pd[2], cdz = pc[2] - pd[2];
    const double bdxcdy = bdx * cdy, cdxbdy = cdx * bdy, cdxady = cdx * ady, adxcdy = adx * cdy, adxbdy = adx * bdy, bdxady = bdx * ady;
    const double det = adz * (bdxcdy - cdxbdy) + bdz * (cdxady - adxcdy) + cdz * (adxbdy - bdxady);
    const double permanent = (fabs(bdxcdy) + fabs(cdxbdy)) * fabs(adz) + (fabs(cdxady) + fabs(adxcdy)) * fabs(bdz) + (fabs(adxbdy) + fabs(bdxady)) * fabs(cdz);
    const double eps = 1.1102230246251565e-16; // 2^-53
    const double errbound = (7.0 + 56.0 * eps) * eps * permanent;
    if (det > errbound) return 1;
    if (-det > errbound) return -1;
    return orient3d_exact(pa, pb, pc, pd);
}

} // namespace o3
} // namespace ipcgpu
