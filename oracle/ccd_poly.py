"""ORACLE / TEST INFRASTRUCTURE -- a second, independent statement of the per-pair continuous collision query.

The reference calls CTCD::vertexFaceCTCD / edgeEdgeCTCD / vertexEdgeCTCD / vertexVertexCTCD (Etienne Vouga's CTCD inside
CCD-Wrapper@23907da, cmake/recipes/ccd_wrapper.cmake:9,16 -- a binary dependency that is not under /root/reference; call sites
src/CollisionObject/SelfCollisionHandler.cpp:607,618,649,660,1024-1265) with a thickness eta: "the earliest t in [0, 1] at which the
two primitives, moving on straight lines, come within eta of each other".  The product (and oracle/orc_contact.cpp::accd) answers
that question by conservative additive advancement on the distance.  This file answers it the way the published algorithm does,
with polynomials in t:

  * vertex-vertex   |a - b|^2 - eta^2 <= 0                                        (quadratic)
  * vertex-edge     |(p - a) x (b - a)|^2 - eta^2 |b - a|^2 <= 0  (quartic)  while the foot of p lies inside the segment:
                    (p - a).(b - a) >= 0 and (b - p).(b - a) >= 0                  (quadratics)
  * vertex-face     ((p - a).n)^2 - eta^2 n.n <= 0 with n = (b - a) x (c - a)      (sextic)  while p projects inside the triangle:
                    ((b - a) x (p - a)).n >= 0, ((c - b) x (p - b)).n >= 0, ((a - c) x (p - c)).n >= 0   (quartics);
                    plus the three vertex-edge and three vertex-vertex tests of the triangle's rim
  * edge-edge       ((c - a).n)^2 - eta^2 n.n <= 0 with n = (b - a) x (d - c)      (sextic)  while the common perpendicular meets both
                    segments: 0 <= ((c - a) x (d - c)).n <= n.n and 0 <= ((c - a) x (b - a)).n <= n.n   (quartics);
                    plus the four vertex-edge and four vertex-vertex tests of the segment ends

Every inequality becomes a set of sub-intervals of [0, 1] through the real roots of its polynomial (companion-matrix eigenvalues,
numpy.roots) and a sign test between consecutive roots; the sets of one test are intersected, the earliest point of any test's set
is the time of impact.  The union of the tests is exactly "distance between the closed primitives <= eta", so this and the
advancement compute the same quantity by unrelated means: tools/ccd_oracle_compare.py measures their difference on the candidate
pairs of a contact scene, tests/test_ccd_poly.py pins both on closed-form cases and on the reference's twelve CTCD booleans
(tests/Collisions/CollisionConstraintTests.cpp:18-35, 83-99)."""
import numpy as np
from numpy.polynomial import polynomial as P

ROOT_IMAG_TOL = 1e-9


def _lin(x, v, c):
    """coordinate c of x + t v as ascending coefficients"""
    return np.array([x[c], v[c]], dtype=np.float64)


def _vec(x, v):
    return [_lin(x, v, 0), _lin(x, v, 1), _lin(x, v, 2)]


def _sub(a, b):
    return [P.polysub(a[c], b[c]) for c in range(3)]


def _dot(a, b):
    return P.polyadd(P.polyadd(P.polymul(a[0], b[0]), P.polymul(a[1], b[1])), P.polymul(a[2], b[2]))


def _cross(a, b):
    return [P.polysub(P.polymul(a[1], b[2]), P.polymul(a[2], b[1])),
            P.polysub(P.polymul(a[2], b[0]), P.polymul(a[0], b[2])),
            P.polysub(P.polymul(a[0], b[1]), P.polymul(a[1], b[0]))]


def _intervals(poly, want_nonpositive, t1=1.0):
    """Sub-intervals of [0, t1] on which poly <= 0 (want_nonpositive) resp. poly >= 0, as a list of (lo, hi)."""
    c = np.trim_zeros(np.asarray(poly, dtype=np.float64), "b")
    scale = np.abs(c).max() if c.size else 0.0
    if c.size == 0 or scale == 0.0:
        return [(0.0, t1)]  # identically zero: the inequality holds everywhere
    c = c / scale
    # strip leading coefficients that are round-off relative to the rest (a degenerate configuration lowers the degree)
    while c.size > 1 and abs(c[-1]) < 1e-14:
        c = c[:-1]
    pts = [0.0, t1]
    if c.size > 1:
        r = np.roots(c[::-1])
        for z in r:
            if abs(z.imag) <= ROOT_IMAG_TOL * max(1.0, abs(z.real)) and 0.0 < z.real < t1:
                pts.append(float(z.real))
    pts = sorted(set(pts))
    out = []
    for lo, hi in zip(pts[:-1], pts[1:]):
        mid = 0.5 * (lo + hi)
        val = P.polyval(mid, c)
        ok = (val <= 0.0) if want_nonpositive else (val >= 0.0)
        if ok:
            if out and abs(out[-1][1] - lo) <= 0.0:
                out[-1] = (out[-1][0], hi)
            else:
                out.append((lo, hi))
    return out


def _intersect(a, b):
    out = []
    for lo1, hi1 in a:
        for lo2, hi2 in b:
            lo, hi = max(lo1, lo2), min(hi1, hi2)
            if lo <= hi:
                out.append((lo, hi))
    return sorted(out)


def _earliest(sets):
    cur = sets[0]
    for s in sets[1:]:
        cur = _intersect(cur, s)
        if not cur:
            return None
    return cur[0][0] if cur else None


def vertex_vertex(a, va, b, vb, eta, t1=1.0):
    d = _sub(_vec(a, va), _vec(b, vb))
    return _earliest([_intervals(P.polysub(_dot(d, d), [eta * eta]), True, t1)])


def vertex_edge(p, vp, a, va, b, vb, eta, t1=1.0):
    Pp, A, B = _vec(p, vp), _vec(a, va), _vec(b, vb)
    ab, ap, pb = _sub(B, A), _sub(Pp, A), _sub(B, Pp)
    cr = _cross(ap, ab)
    dist = P.polysub(_dot(cr, cr), eta * eta * _dot(ab, ab))
    return _earliest([_intervals(dist, True, t1), _intervals(_dot(ap, ab), False, t1), _intervals(_dot(pb, ab), False, t1)])


def _real_roots01(poly, t1=1.0):
    c = np.trim_zeros(np.asarray(poly, dtype=np.float64), "b")
    if c.size <= 1:
        return []
    c = c / np.abs(c).max()
    while c.size > 1 and abs(c[-1]) < 1e-14:
        c = c[:-1]
    if c.size <= 1:
        return []
    out = [float(z.real) for z in np.roots(c[::-1]) if abs(z.imag) <= 1e-7 * max(1.0, abs(z.real)) and -1e-12 <= z.real <= t1 + 1e-12]
    return sorted(min(max(r, 0.0), t1) for r in out)


def _first_root_where(h, conds, t1=1.0):
    """eta = 0: the distance polynomial has no interval of non-positive values, only the coplanarity times (roots of h, a cubic); the
    earliest of them at which every side condition holds (closed: >= -tolerance) is the time of impact."""
    for r in _real_roots01(h, t1):
        ok = True
        for q in conds:
            q = np.asarray(q, dtype=np.float64)
            sc = np.abs(q).max() or 1.0
            if P.polyval(r, q) < -1e-9 * sc:
                ok = False
                break
        if ok:
            return r
    return None


def _min_t(ts):
    ts = [t for t in ts if t is not None]
    return min(ts) if ts else None


def vertex_face(p, vp, a, va, b, vb, c, vc, eta, t1=1.0):
    """Earliest t in [0, t1] at which the point comes within eta of the (closed) triangle, or None."""
    Pp, A, B, C = _vec(p, vp), _vec(a, va), _vec(b, vb), _vec(c, vc)
    n = _cross(_sub(B, A), _sub(C, A))
    h = _dot(_sub(Pp, A), n)
    plane = P.polysub(P.polymul(h, h), eta * eta * _dot(n, n))
    inside = [_dot(_cross(_sub(B, A), _sub(Pp, A)), n), _dot(_cross(_sub(C, B), _sub(Pp, B)), n), _dot(_cross(_sub(A, C), _sub(Pp, C)), n)]
    flat = np.abs(_dot(n, n)).max() <= 1e-300  # a triangle without area throughout: only its rim exists
    if eta == 0.0:
        return None if flat else _first_root_where(h, inside, t1)
    t_face = None if flat else _earliest([_intervals(plane, True, t1)] + [_intervals(q, False, t1) for q in inside])
    rim = [vertex_edge(p, vp, a, va, b, vb, eta, t1), vertex_edge(p, vp, b, vb, c, vc, eta, t1), vertex_edge(p, vp, c, vc, a, va, eta, t1),
           vertex_vertex(p, vp, a, va, eta, t1), vertex_vertex(p, vp, b, vb, eta, t1), vertex_vertex(p, vp, c, vc, eta, t1)]
    return _min_t([t_face] + rim)


def edge_edge(a, va, b, vb, c, vc, d, vd, eta, t1=1.0):
    """Earliest t in [0, t1] at which segment ab comes within eta of segment cd, or None."""
    A, B, C, D = _vec(a, va), _vec(b, vb), _vec(c, vc), _vec(d, vd)
    u, w, r = _sub(B, A), _sub(D, C), _sub(C, A)
    n = _cross(u, w)
    nn = _dot(n, n)
    h = _dot(r, n)
    plane = P.polysub(P.polymul(h, h), eta * eta * nn)
    s_num = _dot(_cross(r, w), n)  # parameter on ab times n.n
    t_num = _dot(_cross(r, u), n)  # parameter on cd times n.n
    uu, ww = _dot(u, u), _dot(w, w)
    parallel = np.abs(nn).max() <= 1e-13 * max(np.abs(P.polymul(uu, ww)).max(), 1e-300)  # n = 0 throughout: no common perpendicular
    if eta == 0.0:
        return None if parallel else _first_root_where(h, [s_num, P.polysub(nn, s_num), t_num, P.polysub(nn, t_num)], t1)
    t_mid = None if parallel else _earliest([_intervals(plane, True, t1), _intervals(s_num, False, t1), _intervals(P.polysub(nn, s_num), False, t1),
                       _intervals(t_num, False, t1), _intervals(P.polysub(nn, t_num), False, t1)])
    ends = [vertex_edge(a, va, c, vc, d, vd, eta, t1), vertex_edge(b, vb, c, vc, d, vd, eta, t1),
            vertex_edge(c, vc, a, va, b, vb, eta, t1), vertex_edge(d, vd, a, va, b, vb, eta, t1),
            vertex_vertex(a, va, c, vc, eta, t1), vertex_vertex(a, va, d, vd, eta, t1),
            vertex_vertex(b, vb, c, vc, eta, t1), vertex_vertex(b, vb, d, vd, eta, t1)]
    return _min_t([t_mid] + ends)


def toi(kind, X, V, eta, t1=1.0):
    """kind 2: point-triangle (X[0] against X[1..3]); 3: edge-edge (X[0]X[1] against X[2]X[3]).  X, V: 4 x 3.  None = no contact."""
    X, V = np.asarray(X, dtype=np.float64), np.asarray(V, dtype=np.float64)
    if kind == 2:
        return vertex_face(X[0], V[0], X[1], V[1], X[2], V[2], X[3], V[3], eta, t1)
    return edge_edge(X[0], V[0], X[1], V[1], X[2], V[2], X[3], V[3], eta, t1)
