"""ipc_amd/csrc/orient3d_exact.h -- the exact orientation predicate behind the USE_PREDICATES variant of the intersection checks
(IglUtils.hpp:222-233, 280-294) -- compiled for the host and pinned on exact rational arithmetic: sign det [a - d; b - d; c - d] with
fractions.Fraction, on generic, nearly coplanar and exactly coplanar configurations."""
import ctypes
import os
import subprocess
from fractions import Fraction

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def o3():
    out = os.path.join(HERE, "orient3d", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libo3.so")
    src = os.path.join(HERE, "orient3d", "o3_host.cpp")
    hdr = os.path.join(HERE, "..", "ipc_amd", "csrc", "orient3d_exact.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-shared", "-fPIC", src, "-o", so])
    lib = ctypes.CDLL(so)
    for f in (lib.o3_orient3d, lib.o3_orient3d_exact):
        f.argtypes = [ctypes.c_void_p]
        f.restype = ctypes.c_int
    return lib


def exact_sign(P):
    a, b, c, d = [[Fraction(float(x)) for x in p] for p in P]
    m = [[a[k] - d[k] for k in range(3)], [b[k] - d[k] for k in range(3)], [c[k] - d[k] for k in range(3)]]
    det = (m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0])
           + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]))
    return (det > 0) - (det < 0)


def call(f, P):
    P = np.ascontiguousarray(P, dtype=np.float64)
    return f(P.ctypes.data)


def test_generic_configurations(o3):
    rng = np.random.default_rng(1)
    for _ in range(300):
        P = rng.normal(size=(4, 3)) * 10.0 ** rng.integers(-3, 4)
        assert call(o3.o3_orient3d, P) == exact_sign(P) == call(o3.o3_orient3d_exact, P)


def test_nearly_coplanar_configurations(o3):
    """d in the plane of a, b, c up to the last bits: the floating-point determinant is noise, the filter must hand over to the exact path"""
    rng = np.random.default_rng(2)
    seen = {-1: 0, 0: 0, 1: 0}
    for _ in range(600):
        a, b, c = rng.normal(size=(3, 3)) * 10.0 ** rng.integers(-2, 3)
        s, t = rng.uniform(-1, 2, size=2)
        d = a + s * (b - a) + t * (c - a)
        d = d + rng.integers(-2, 3, size=3) * np.spacing(d)  # a few units in the last place off the plane
        P = np.array([a, b, c, d])
        want = exact_sign(P)
        seen[want] += 1
        assert call(o3.o3_orient3d, P) == want and call(o3.o3_orient3d_exact, P) == want
    assert seen[1] > 100 and seen[-1] > 100


def test_exactly_coplanar_and_degenerate(o3):
    rng = np.random.default_rng(3)
    for _ in range(200):
        a, b, c = rng.integers(-50, 50, size=(3, 3)).astype(float)
        i, j = rng.integers(-3, 4, size=2)
        d = a + i * (b - a) + j * (c - a)  # integers: exactly in the plane
        sc = 2.0 ** rng.integers(-20, 20)
        P = np.array([a, b, c, d]) * sc
        assert exact_sign(P) == 0 and call(o3.o3_orient3d, P) == 0
    P = np.zeros((4, 3))
    assert call(o3.o3_orient3d, P) == 0
    P = np.array([[1e300, 0, 0], [0, 1e-300, 0], [0, 0, 1.0], [0, 0, 0]])  # products that overflow / underflow nowhere: sign +1
    assert call(o3.o3_orient3d, P) == exact_sign(P)


def test_translation_far_from_the_origin(o3):
    """coordinates of 1e8 with features of 1e-6: a - d is rounded, the exact path works on the original coordinates"""
    rng = np.random.default_rng(4)
    for _ in range(300):
        base = rng.normal(size=3) * 1e8
        P = base + rng.normal(size=(4, 3)) * 1e-6
        P[3] = P[0] + rng.uniform(-1, 2) * (P[1] - P[0]) + rng.uniform(-1, 2) * (P[2] - P[0])
        assert call(o3.o3_orient3d, P) == exact_sign(P)


def seg_tri_exact_python(X):
    """IglUtils::segTriIntersect, USE_PREDICATES branch (IglUtils.hpp:222-233, 246-264): exact plane-side test, then the floating-point solve"""
    ve0, ve1, vt0, vt1, vt2 = [np.asarray(p, dtype=np.float64) for p in X]
    o1, o2 = exact_sign([vt0, vt1, vt2, ve0]), exact_sign([vt0, vt1, vt2, ve1])
    if o1 == 0 or o2 == 0 or o1 == o2:
        return False
    # (u, v, t) of [vt1 - vt0, vt2 - vt0, ve0 - ve1] (u v t)^T = ve0 - vt0 in rational arithmetic; None when the floating-point solve that
    # follows in the reference (fullPivLu) decides a bound by its rounding -- those cases pin nothing
    F = lambda p: [Fraction(float(x)) for x in p]  # noqa: E731
    a, b, c, r = [F(q) for q in (vt1 - vt0, vt2 - vt0, ve0 - ve1, ve0 - vt0)]
    det3 = lambda x, y, z: (x[0] * (y[1] * z[2] - y[2] * z[1]) - y[0] * (x[1] * z[2] - x[2] * z[1]) + z[0] * (x[1] * y[2] - x[2] * y[1]))  # noqa: E731
    D = det3(a, b, c)
    if D == 0:
        return False
    u, v, t = det3(r, b, c) / D, det3(a, r, c) / D, det3(a, b, r) / D
    margins = [u, v, 1 - u - v, t, 1 - t]
    if min(abs(float(m)) for m in margins) < 1e-9:
        return None
    return all(m >= 0 for m in margins)


def seg_tri_cases(rng, n):
    out = []
    for _ in range(n):
        tri = rng.normal(size=(3, 3))
        kind = rng.integers(0, 4)
        if kind == 0:  # generic
            e = rng.normal(size=(2, 3))
        elif kind == 1:  # through the interior
            c = tri.mean(0)
            d = rng.normal(size=3)
            e = np.array([c + d, c - rng.uniform(0.1, 2) * d])
        elif kind == 2:  # one end IN the plane (to the last bits): not an intersection in the exact branch, a coin toss without it
            s, t = rng.uniform(0, 0.5, size=2)
            p = tri[0] + s * (tri[1] - tri[0]) + t * (tri[2] - tri[0])
            p = p + rng.integers(-1, 2, size=3) * np.spacing(p)
            e = np.array([p, p + rng.normal(size=3)])
        else:  # both ends on one side, close to the plane
            nrm = np.cross(tri[1] - tri[0], tri[2] - tri[0])
            c = tri.mean(0)
            e = np.array([c + 1e-13 * nrm + 0.1 * (tri[1] - tri[0]), c + 2e-13 * nrm - 0.1 * (tri[2] - tri[0])])
        out.append(np.vstack([e, tri]))
    return out


def test_oracle_segment_triangle_with_exact_predicates():
    """oracle/orc_contact.cpp::segTriIntersect(exact = true) beside the statement above"""
    from oracle import orc
    L = orc.lib()
    L.orc_seg_tri_intersect_exact.argtypes = [ctypes.c_void_p]
    L.orc_seg_tri_intersect_exact.restype = ctypes.c_int
    rng = np.random.default_rng(9)
    hits = 0
    for X in seg_tri_cases(rng, 800):
        Xc = np.ascontiguousarray(X, dtype=np.float64)
        want = seg_tri_exact_python(X)
        if want is None:
            continue
        hits += want
        assert bool(L.orc_seg_tri_intersect_exact(Xc.ctypes.data)) == want
    assert 100 < hits < 700
