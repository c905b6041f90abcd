"""Parity of the HIP contact path (C ABI) against the CPU oracle: surface bookkeeping and constraint sets bit-exact
(integers), barrier energy / gradient 1e-10 relative, PSD-projected barrier Hessian entries 1e-9 relative."""
import numpy as np
import pytest

from ipc_amd import scene

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def two_blocks(gap, n=4, shift=0.13):
    Va, Fa = scene.make_box(n, 1, n, size=(1.0, 0.3, 1.0), origin=(0, 0, 0))
    Vb, Fb = scene.make_box(n, 1, n, size=(1.0, 0.3, 1.0), origin=(shift, 0.3 + gap, 0.5 * shift))
    return np.vstack([Va, Vb]), np.vstack([Fa, Fb + Va.shape[0]])


@pytest.fixture(scope="module")
def pair(orc, gpu_lib):
    V, F = two_blocks(0.004)
    V = scene.jitter(V, F, rel=3e-3)
    SF = scene.surface_tris(F)
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_surface(SF)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    c.opt_init(0.025, False)
    c.set_surface(SF)
    dbc = np.nonzero(V[:, 1] < 1e-3 + V[:, 1].min())[0].astype(np.int32)  # bottom face of the lower slab
    m.set_dbc(dbc, 1)
    c.set_dbc(dbc, 1)
    dHat = 1e-3 ** 2 * m.features()["bboxDiag2"] * 40
    return dict(V=V, F=F, SF=SF, m=m, c=c, dHat=dHat, dbc=dbc)


def test_surface_bookkeeping_bit_exact(orc, pair):
    svi_o, sfe_o = orc.mesh_surface(pair["m"])
    svi_g, sfe_g = pair["c"].get_surface()
    assert np.array_equal(svi_g, svi_o) and np.array_equal(sfe_g, sfe_o)


def test_constraint_sets_bit_exact(orc, pair):
    o = orc.Contacts().build(pair["m"], pair["dHat"], brute=True)
    g = pair["c"].contact_build(pair["dHat"])
    assert len(o["active"]) > 30
    for k in ("active", "para", "para_eiej", "cs_ptee"):
        assert np.array_equal(g[k], o[k]), k  # same tuples in the same order
    # a wider activation distance: more pairs, PP/PE duplicates with multiplicities
    o2 = orc.Contacts().build(pair["m"], 4 * pair["dHat"])
    g2 = pair["c"].contact_build(4 * pair["dHat"])
    for k in ("active", "para", "para_eiej", "cs_ptee"):
        assert np.array_equal(g2[k], o2[k]), k
    assert len(g2["active"]) >= len(g["active"])
    o3 = orc.Contacts().build(pair["m"], 0.12 * pair["dHat"])
    g3 = pair["c"].contact_build(0.12 * pair["dHat"])
    for k in ("active", "para", "para_eiej", "cs_ptee"):
        assert np.array_equal(g3[k], o3[k]), k
    assert 0 < len(g3["active"]) < len(g["active"])
    # empty set far away
    assert len(pair["c"].contact_build(1e-12)["active"]) == 0
    pair["c"].contact_build(pair["dHat"])


def test_nearly_parallel_edges_go_to_the_mollified_set(orc, gpu_lib):
    # unjittered slabs: facing edges are exactly parallel -> paraEE entries with their (eI, eJ) bookkeeping
    V, F = two_blocks(0.004, n=3, shift=0.0)
    SF = scene.surface_tris(F)
    m = orc.Mesh(V, F)
    m.set_surface(SF)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F)
    c.opt_init(0.025, False)
    c.set_surface(SF)
    dHat = (0.006) ** 2
    o = orc.Contacts().build(m, dHat)
    g = c.contact_build(dHat)
    assert len(o["para"]) > 0
    for k in ("active", "para", "para_eiej", "cs_ptee"):
        assert np.array_equal(g[k], o[k]), k
    cs = orc.Contacts()
    cs.build(m, dHat)
    assert abs(c.contact_energy(dHat, 1e3) - cs.energy(m, dHat, 1e3)) <= 1e-10 * abs(cs.energy(m, dHat, 1e3))
    assert relerr(c.contact_gradient_add(dHat, 1e3, True), cs.gradient(m, dHat, 1e3, True)) < 1e-10
    pairs = c.contact_connectivity()
    assert np.array_equal(pairs, cs.connectivity(m))
    c.set_pattern(pairs)
    m2 = orc.Mesh(V, F)
    m2.set_surface(SF)
    ia, ja = m2.pattern(extra_edges=pairs)
    c.set_zero()
    c.contact_hessian_add(dHat, 1e3, True)
    assert relerr(c.get_a(), cs.hessian(m2, len(ja), dHat, 1e3, True)) < 1e-9
    c.close()


def test_ccd_step_bounds_and_intersection_check(orc, pair):
    """'Bit-exact CCD index': the limiting pair is the same tuple, the bound agrees to round-off."""
    m, c, dHat, V = pair["m"], pair["c"], pair["dHat"], pair["V"]
    nA = V.shape[0] // 2
    cs = orc.Contacts()
    cs.build(m, dHat)
    cand = cs.get()["cs_ptee"]
    c.contact_build(dHat)
    rng = np.random.default_rng(21)
    for trial in range(4):
        p = 0.002 * rng.normal(size=V.shape)
        p[nA:, 1] -= 0.02 * (trial + 1)  # upper slab pushed into the lower one
        so, arg = orc.ccd_partial(cs, m, p.reshape(-1), 0.8, 1.0)
        sg, pg = c.ccd_partial(p.reshape(-1), 0.8, 1.0)
        assert abs(sg - so) <= 1e-12 * so and so < 1.0
        assert pg == tuple(int(x) for x in cand[arg])
        fo, pfo, no = orc.ccd_full(m, p.reshape(-1), 0.8, 1.0)
        fg, pfg, ng = c.ccd_full(p.reshape(-1), 0.8, 1.0)
        assert abs(fg - fo) <= 1e-12 * fo and pfg == pfo and ng == no
        assert fo <= so
        # stepping by the bound keeps the mesh intersection-free, the full step does not
        c.set_positions(V + fg * p)
        assert not c.is_intersected()
        c.set_positions(V + p)
        m.set_V(V + p)
        assert c.is_intersected() == orc.is_intersected(m)
        c.set_positions(V)
        m.set_V(V)
    # separating motion: no pair limits the step
    p = np.zeros_like(V)
    p[nA:, 1] = 0.05
    assert c.ccd_full(p.reshape(-1), 0.8, 0.7)[0] == 0.7 and c.ccd_partial(p.reshape(-1), 0.8, 0.7)[0] == 0.7
    assert not c.is_intersected()


def test_grid_over_the_last_measured_box_survives_a_mesh_that_moves_away(orc, pair):
    """Constraint-set build and intersection check lay their grid over the bounding box the LAST build or check measured and read everything back with one
    synchronisation (round 6).  A mesh that has left that box is noticed on the device (stale flag: nothing runs) and the pass repeats on the fresh box;
    a state with many more cell entries than the last one grows the lists and repeats.  Same answers as the oracle either way."""
    m, c, dHat, V = pair["m"], pair["c"], pair["dHat"], pair["V"]
    nA = V.shape[0] // 2
    assert len(c.contact_build(dHat)["active"]) > 30  # (measures the box at V)
    for shift in ([7.0, -3.0, 11.0], [-40.0, 0.5, 0.25], [0.0, 0.0, 0.0]):  # far outside the last box, again, and back
        X = V + np.asarray(shift)
        c.set_positions(X)
        m.set_V(X)
        assert c.is_intersected() == orc.is_intersected(m) == False  # noqa: E712
        got = c.contact_build(dHat)
        want = orc.Contacts().build(m, dHat)
        assert len(want["active"]) > 30
        for k in ("active", "para", "para_eiej", "cs_ptee"):
            assert np.array_equal(got[k], want[k]), (shift, k)
        # pierced state at the shifted place: the upper slab pushed through the lower one
        P = X.copy()
        P[nA:, 1] -= 0.06
        c.set_positions(P)
        m.set_V(P)
        assert c.is_intersected() == orc.is_intersected(m) == True  # noqa: E712
    c.set_positions(V)
    m.set_V(V)


def drop_scene(orc, gpu_lib, n=2, gap=0.03, dHatEps=1e-2, dt=0.01, speed=-1.5, jitter=1e-2):
    """Two slabs, the upper one falling on the clamped lower one.  The start state is a jittered (pre-strained) copy of the
    rest shape: exactly at rest the reference's makePD2d is discontinuous (see test_gpu_parity), which would make an
    iterate-by-iterate comparison depend on round-off."""
    V, F = two_blocks(gap, n=n)
    Vstart = scene.jitter(V, F, rel=jitter) if jitter else V
    SF = scene.surface_tris(F)
    nA = V.shape[0] // 2
    bottom = np.nonzero(V[:nA, 1] < V[:nA, 1].min() + 0.02)[0].astype(np.int32)
    vel = np.zeros_like(V)
    vel[nA:, 1] = speed
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_surface(SF)
    m.set_dbc(bottom, 1)
    m.set_V(Vstart)
    o = orc.Optimizer(m, dt=dt, gravity=True, nthreads=4)
    orc.opt_enable_self_collision(o, dHatEps)
    orc.opt_set_velocity(o, vel)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    c.set_dbc(bottom, 1)
    c.set_positions(Vstart)
    c.opt_init(dt, True)
    c.set_surface(SF)
    c.enable_self_collision(dHatEps)
    c.set_velocity(vel)
    return m, o, c, nA


@pytest.mark.parametrize("n,speed,dt", [(2, -1.5, 0.01), (3, -6.0, 0.02)])
def test_contact_newton_iterates_track_the_oracle(orc, gpu_lib, n, speed, dt):
    """The whole contact-aware stepper (constraint sets per trial, barrier terms, adaptive kappa, partial / CFL / full CCD,
    intersection-checked line search, pattern growth + re-analysis) iterate by iterate against the oracle."""
    m, o, c, nA = drop_scene(orc, gpu_lib, n=n, speed=speed, dt=dt)
    o.precompute()
    c.precompute()
    seen, worst = 0, 0.0
    for step in range(6):
        o.begin_timestep()
        c.begin_timestep()
        so, sg = o.state(), c.state()
        assert sg["dHat"] == so["dHat"]
        assert abs(sg["kappa"] - so["kappa"]) <= 1e-9 * so["kappa"]
        assert abs(sg["E"] - so["E"]) <= 1e-9 * abs(so["E"])
        for it in range(60):
            co, cg = o.newton_iter(), c.newton_iter()
            assert co == cg, (step, it)
            if co:
                break
            so, sg = o.state(), c.state()
            cst_o, cst_g = orc.opt_contact_state(o), c.contact_state()
            assert cst_g["nActive"] == len(cst_o["active"]) and cst_g["nPara"] == len(cst_o["para"]), (step, it)
            assert abs(sg["alphaFeasible"] - so["alphaFeasible"]) <= 1e-8 * so["alphaFeasible"], (step, it)
            assert sg["stepSize"] == pytest.approx(so["stepSize"], rel=1e-8), (step, it)
            assert abs(sg["kappa"] - so["kappa"]) <= 1e-8 * so["kappa"], (step, it)
            assert abs(sg["E"] - so["E"]) <= 1e-8 * abs(so["E"]), (step, it)
            worst = max(worst, relerr(sg["V"], so["V"]))
            seen = max(seen, cst_g["nActive"])
        else:
            pytest.fail("contact Newton did not converge")
        o.end_timestep()
        c.end_timestep()
        assert c.check_inversion() and not c.is_intersected()
    assert seen > 0 and c.contact_state()["nPatternChanges"] >= 1
    assert c.contact_state()["nFullCCD"] == orc.opt_contact_state(o)["n_full_ccd"]
    assert worst < 1e-8
    Vn = c.state()["V"]
    assert Vn[nA:, 1].mean() > Vn[:nA, 1].mean() + 0.2
    c.close()


def test_barrier_energy_gradient_hessian(orc, pair):
    m, c, dHat = pair["m"], pair["c"], pair["dHat"]
    cs = orc.Contacts()
    sets = cs.build(m, dHat)
    c.contact_build(dHat)
    for kappa in (1.0, 2.5e4):
        Eo = cs.energy(m, dHat, kappa)
        assert abs(c.contact_energy(dHat, kappa) - Eo) <= 1e-10 * abs(Eo)
        for proj in (True, False):
            go = cs.gradient(m, dHat, kappa, proj)
            gg = c.contact_gradient_add(dHat, kappa, proj)
            assert relerr(gg, go) < 1e-10
            if proj:
                for v in pair["dbc"]:
                    assert np.all(gg[3 * v:3 * v + 3] == 0)
    # hand the same set in through the ABI (an adapter that keeps the reference's own constraint code)
    c.contact_set(sets["active"], sets["para"], sets["para_eiej"])
    assert abs(c.contact_energy(dHat, 7.0) - cs.energy(m, dHat, 7.0)) <= 1e-10 * abs(cs.energy(m, dHat, 7.0))
    # Hessian: the pattern must contain the contact connectivity
    pairs = c.contact_connectivity()
    assert np.array_equal(pairs, cs.connectivity(m))
    c.set_pattern()
    with pytest.raises(Exception):
        c.set_zero()
        c.contact_hessian_add(dHat, 1e3, True)
    c.set_pattern(pairs)
    m2 = orc.Mesh(pair["V"], pair["F"], YM=1e5, PR=0.4, density=1000.0)
    m2.set_surface(pair["SF"])
    m2.set_dbc(pair["dbc"], 1)
    ia, ja = m2.pattern(extra_edges=pairs)
    ia_g, ja_g = c.get_pattern()
    assert np.array_equal(ia_g, ia) and np.array_equal(ja_g, ja)
    for proj in (True, False):
        c.set_zero()
        c.contact_hessian_add(dHat, 1e3, proj)
        a_o = cs.hessian(m2, len(ja), dHat, 1e3, proj)
        a_g = c.get_a()
        assert relerr(a_g, a_o) < 1e-9
        assert np.array_equal(a_g == 0, a_o == 0)
    # elastic + barrier in one matrix still factorises (SPD) and solves
    c.assemble_newton(0.025 ** 2, True, with_gradient=False)
    c.contact_hessian_add(dHat, 1e3, True)
    c.analyze_pattern()
    assert c.factorize()
    b = np.random.default_rng(2).normal(size=len(ia) - 1)
    x = c.solve(b)
    assert np.linalg.norm(c.multiply(x) - b) <= 1e-9 * np.linalg.norm(b)


# ---- analytic half-space -------------------------------------------------------------------------
def tilted_block(n=2, lift=0.004):
    V, F = scene.make_box(n, n, n, size=(0.5, 0.5, 0.5), origin=(0, 0, 0))
    nrm = np.array([0.1, 1.0, -0.05])
    nrm /= np.linalg.norm(nrm)
    Vs = scene.jitter(V, F, rel=1e-2)
    origin = nrm * ((Vs @ nrm).min() - lift)
    return V, Vs, F, origin, nrm


def test_half_space_pieces(orc, gpu_lib):
    V, Vs, F, origin, nrm = tilted_block(n=3)
    SF = scene.surface_tris(F)
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_surface(SF)
    m.set_V(Vs)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    c.set_positions(Vs)
    c.opt_init(0.01, True)
    c.set_surface(SF)
    idx = c.add_half_space(origin, 2.5 * nrm, 1e-2)
    hs = orc.HalfSpace(origin, 2.5 * nrm)
    dist = Vs @ nrm - origin @ nrm
    dHat = (np.sort(dist)[9] * 1.0001) ** 2
    vo, vg = hs.build(m, dHat), c.halfspace_build(idx, dHat)
    assert len(vo) >= 9 and np.array_equal(vg, vo)  # integer output: bit-exact, same order
    some = vo[:3]
    m.set_dbc(some, 1)
    c.set_dbc(some, 1)
    assert np.array_equal(c.halfspace_build(idx, dHat), hs.build(m, dHat))
    m.clear_dbc()
    c.clear_dbc()
    hs.build(m, dHat)
    c.halfspace_build(idx, dHat)
    kappa = 4.2e3
    Eo = hs.energy(m, dHat, kappa)
    assert abs(c.halfspace_energy(idx, dHat, kappa) - Eo) <= 1e-12 * abs(Eo)
    assert relerr(c.halfspace_gradient_add(idx, dHat, kappa), hs.gradient(m, dHat, kappa)) < 1e-12
    c.set_pattern()
    ia, ja = m.pattern()
    c.set_zero()
    c.halfspace_hessian_add(idx, dHat, kappa, True)
    a_o = hs.hessian(m, len(ja), dHat, kappa, True)
    assert relerr(c.get_a(), a_o) < 1e-12 and np.count_nonzero(a_o) > 0
    rng = np.random.default_rng(4)
    for trial in range(3):
        p = 0.05 * rng.normal(size=V.shape)
        p[:, 1] -= 0.02 * trial
        so, sg = hs.step_bound(m, p.reshape(-1), 0.9, 1.0), c.halfspace_step_bound(idx, p.reshape(-1), 0.9, 1.0)
        assert abs(sg - so) <= 1e-14 * abs(so)
    # HalfSpace::move (ipcgpu_halfspace_move): towards the block the nearest surface node cuts the move short, away from it the whole move is taken;
    # the plane the library then uses is the moved one (same active set, same energy as the moved oracle plane)
    for delta in (0.5 * nrm, 0.003 * nrm + np.array([0.02, 0.0, -0.01]), -0.01 * nrm):
        (o_new, left_o), left_g = hs.move(m, delta, 0.5), c.half_space_move(idx, delta, 0.5)
        assert abs(left_g - left_o) <= 1e-14, (left_g, left_o)
        vo2 = hs.build(m, dHat)
        assert np.array_equal(c.halfspace_build(idx, dHat), vo2)
        Eo2 = hs.energy(m, dHat, kappa)
        assert abs(c.halfspace_energy(idx, dHat, kappa) - Eo2) <= 1e-11 * max(abs(Eo2), 1e-30)
    assert left_o == 0.0
    # a set handed in through the ABI
    c.halfspace_set(idx, vo[::2])
    hs2 = orc.HalfSpace(origin, nrm)
    big = hs2.build(m, 1e9)
    assert len(big) == len(np.unique(SF))
    c.close()


def test_half_space_newton_iterates_track_the_oracle(orc, gpu_lib):
    """A block dropped on a tilted plane: half-space constraint set, barrier terms, kappa adaptation and the ray step bound
    inside the stepper, iterate by iterate."""
    V, Vs, F, origin, nrm = tilted_block(n=2, lift=0.03)
    SF = scene.surface_tris(F)
    vel = np.zeros_like(V)
    vel[:, 1] = -2.0
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_surface(SF)
    m.set_V(Vs)
    o = orc.Optimizer(m, dt=0.01, gravity=True, nthreads=2)
    orc.opt_add_half_space(o, origin, nrm, 1e-2)
    orc.opt_set_velocity(o, vel)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    c.set_positions(Vs)
    c.opt_init(0.01, True)
    c.set_surface(SF)
    c.add_half_space(origin, nrm, 1e-2)
    c.set_velocity(vel)
    o.precompute()
    c.precompute()
    touched, limited = 0, 0
    for step in range(8):
        o.begin_timestep()
        c.begin_timestep()
        assert abs(c.state()["kappa"] - o.state()["kappa"]) <= 1e-9 * o.state()["kappa"]
        for it in range(60):
            co, cg = o.newton_iter(), c.newton_iter()
            assert co == cg, (step, it)
            if co:
                break
            so, sg = o.state(), c.state()
            assert c.contact_state()["nHalfSpace"] == len(orc.opt_half_space_set(o)), (step, it)
            assert abs(sg["alphaFeasible"] - so["alphaFeasible"]) <= 1e-9 * so["alphaFeasible"], (step, it)
            assert abs(sg["stepSize"] - so["stepSize"]) <= 1e-9 * so["stepSize"], (step, it)
            assert abs(sg["kappa"] - so["kappa"]) <= 1e-9 * so["kappa"], (step, it)
            assert abs(sg["E"] - so["E"]) <= 1e-9 * abs(so["E"]), (step, it)
            assert relerr(sg["V"], so["V"]) < 1e-9, (step, it)
            limited += so["alphaFeasible"] < 1.0
        else:
            pytest.fail("Newton did not converge")
        o.end_timestep()
        c.end_timestep()
        touched = max(touched, c.contact_state()["nHalfSpace"])
        assert ((c.state()["V"] - origin) @ nrm).min() > 0
    assert touched > 0 and limited > 0
    c.close()


def test_mat_stack_sets_and_iterates_at_moderate_size(orc, gpu_lib):
    """SURVEY 8(d) config 5 in small: two mat sheets closer than sqrt(dHat) -- thousands of active pairs, a broad-phase grid
    with many primitives per cell, duplicate PP / PE merges -- sets bit-exact, then the stepper iterate by iterate."""
    V, F, nA = scene.make_mat_stack(14, 2, gap=1.2e-3)
    Vs = scene.jitter(V, F, rel=2e-3)
    SF = scene.surface_tris(F)
    border = np.nonzero((np.abs(V[:nA, 0]) > 0.49) | (np.abs(V[:nA, 2]) > 0.49))[0].astype(np.int32)
    vel = np.zeros_like(V)
    vel[nA:, 1] = -0.05
    m = orc.Mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
    m.set_surface(SF)
    m.set_dbc(border, 1)
    m.set_V(Vs)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
    c.set_dbc(border, 1)
    c.set_positions(Vs)
    c.opt_init(0.01, True)
    c.set_surface(SF)
    dHat = 1e-6 * m.features()["bboxDiag2"]
    o_sets = orc.Contacts().build(m, dHat)
    g_sets = c.contact_build(dHat)
    assert len(o_sets["active"]) > 1000
    for k in ("active", "para", "para_eiej", "cs_ptee"):
        assert np.array_equal(g_sets[k], o_sets[k]), k
    o = orc.Optimizer(m, dt=0.01, gravity=True, nthreads=4)
    orc.opt_enable_self_collision(o, 1e-3)
    orc.opt_set_velocity(o, vel)
    c.enable_self_collision(1e-3)
    c.set_velocity(vel)
    o.precompute()
    c.precompute()
    for step in range(2):
        o.begin_timestep()
        c.begin_timestep()
        for it in range(30):
            co, cg = o.newton_iter(), c.newton_iter()
            assert co == cg, (step, it)
            if co:
                break
            so, sg = o.state(), c.state()
            assert c.contact_state()["nActive"] == len(orc.opt_contact_state(o)["active"]), (step, it)
            assert abs(sg["stepSize"] - so["stepSize"]) <= 1e-9 * so["stepSize"], (step, it)
            assert abs(sg["E"] - so["E"]) <= 1e-10 * abs(so["E"]), (step, it)
            assert relerr(sg["V"], so["V"]) < 1e-10, (step, it)
        o.end_timestep()
        c.end_timestep()
    assert not c.is_intersected()
    c.close()


def test_contact_assembly_is_bit_reproducible(gpu_lib):
    """The barrier / friction forces and Hessian blocks are scattered by a sorted segmented reduction (hip_contact.hip "deterministic
    scatter"), not by fp64 atomics: evaluated twice at the same state they give the SAME BITS, and so does the elastic pass (gradient and
    CSR values)."""
    V, F, nA = scene.make_mat_stack(14, 2, gap=1.2e-3)
    Vs = scene.jitter(V, F, rel=2e-3)
    SF = scene.surface_tris(F)
    border = np.nonzero((np.abs(V[:nA, 0]) > 0.49) | (np.abs(V[:nA, 2]) > 0.49))[0].astype(np.int32)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
    c.set_dbc(border, 1)
    c.set_positions(Vs)
    c.opt_init(0.01, True)
    c.set_surface(SF)
    dHat = 1e-6 * float(np.sum((V.max(0) - V.min(0)) ** 2))
    sets = c.contact_build(dHat)
    assert len(sets["active"]) > 1000
    c.set_pattern(c.contact_connectivity())
    report = {}
    for name, fn in (("barrier gradient", lambda: c.contact_gradient_add(dHat, 1e3, True)),
                     ("elastic gradient", lambda: c.assemble_newton(1e-4, True)),
                     ("barrier Hessian", lambda: (c.set_zero(), c.contact_hessian_add(dHat, 1e3, True), c.get_a().copy())[2]),
                     ("elastic Hessian", lambda: (c.assemble_newton(1e-4, True, with_gradient=False), c.get_a().copy())[1])):
        runs = [np.array(fn(), copy=True) for _ in range(4)]
        report[name] = max(float(np.abs(r - runs[0]).max()) for r in runs[1:])
    c.close()
    assert report["barrier gradient"] == 0.0 and report["barrier Hessian"] == 0.0 and report["elastic Hessian"] == 0.0, report
    assert report["elastic gradient"] == 0.0, report


def test_codimensional_points_and_segments_bookkeeping_sets_and_intersection(orc, gpu_lib):
    """ipcgpu_set_surface_codim (Mesh.cpp:490-515, 912-920): SVI / SFEdges equal to the oracle's with a segment and isolated points in the
    mesh; the constraint sets they take part in; k_points_in_tets (SelfCollisionHandler.cpp:3301-3338) beside the oracle's loop."""
    from test_oracle_contact import codim_point_mesh
    Vall, F, SF, CE, n = codim_point_mesh()
    m = orc.Mesh(Vall, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_surface(SF, CE)
    c = gpu_lib.Context(0)
    c.set_mesh(Vall, F, YM=1e5, PR=0.4, density=1000.0)
    c.set_surface(SF, CE)
    svi_o, sfe_o = orc.mesh_surface(m)
    svi_g, sfe_g = c.get_surface()
    assert np.array_equal(svi_o, svi_g) and np.array_equal(np.asarray(sfe_o).reshape(-1, 2), sfe_g)
    V2 = Vall.copy()
    for pos, want in (([0.31, 0.47, 0.52], True), ([0.5, 1.3, 0.5], False), ([0.5, 0.999, 0.5], True), ([0.25, 0.25, 1e-9], True), ([0.5, 0.5, -1e-9], False)):
        V2[n] = pos
        m.set_V(V2)
        c.set_positions(V2)
        assert orc.is_intersected(m) == want and c.is_intersected() == want, pos
    # a point hovering over the top face and the segment lowered next to an edge of the box: the same constraint tuples
    V2[n] = [0.4, 1.0 + 2e-3, 0.6]
    V2[n + 3] = [-0.2, 1.0 + 1e-3, 0.3]
    V2[n + 4] = [1.2, 1.0 + 1e-3, 0.35]
    m.set_V(V2)
    c.set_positions(V2)
    dHat = 1e-4
    o = orc.Contacts().build(m, dHat)
    g = c.contact_build(dHat)
    assert len(o["active"]) > 0
    for k in ("active", "para", "para_eiej", "cs_ptee"):
        assert np.array_equal(g[k], o[k]), k
    assert any(int(t[0]) == -n - 1 for t in np.asarray(o["active"]).reshape(-1, 4))  # the point is the vertex of a PT / PE / PP tuple
    c.close()


def test_intersection_checks_with_exact_predicates(orc, gpu_lib):
    """ipcgpu_set_exact_predicates: the plane-side tests of the intersection checks as a USE_PREDICATES build of the reference makes them
    (IglUtils.hpp:222-233, 280-294; orient3d_exact.h pinned on rational arithmetic in tests/test_orient3d.py).  A segment that ends exactly IN
    the plane of a face touches it in the default build and does not in this one; a point exactly ON a face of a tetrahedron is inside in both."""
    from test_oracle_contact import codim_point_mesh
    Vall, F, SF, CE, n = codim_point_mesh()
    m = orc.Mesh(Vall, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_surface(SF, CE)
    c = gpu_lib.Context(0)
    c.set_mesh(Vall, F, YM=1e5, PR=0.4, density=1000.0)
    c.set_surface(SF, CE)
    rng = np.random.default_rng(12)
    V2 = Vall.copy()
    V2[n] = [5.0, 5.0, 5.0]  # the points out of the way
    cases = [([0.3, 1.0, 0.4], [0.35, 1.6, 0.45], True, False),  # one end exactly in the plane y = 1 of the top face
             ([0.3, 0.9, 0.4], [0.35, 1.6, 0.45], True, True),  # through the top face
             ([0.3, 1.0 + 1e-15, 0.4], [0.35, 1.6, 0.45], False, False)]  # a hair above it
    for e0, e1, want_default, want_exact in cases:
        V2[n + 3], V2[n + 4] = e0, e1
        m.set_V(V2)
        c.set_positions(V2)
        for exact, want in ((False, want_default), (True, want_exact)):
            m.set_exact_predicates(exact)
            c.set_exact_predicates(exact)
            assert orc.is_intersected(m) == want and c.is_intersected() == want, (e0, exact)
    # random segments around the box, both modes: oracle and GPU agree
    agree = 0
    for trial in range(40):
        V2[n + 3] = rng.uniform(-0.3, 1.3, size=3)
        V2[n + 4] = rng.uniform(-0.3, 1.3, size=3)
        m.set_V(V2)
        c.set_positions(V2)
        for exact in (False, True):
            m.set_exact_predicates(exact)
            c.set_exact_predicates(exact)
            assert orc.is_intersected(m) == c.is_intersected()
            agree += 1
    # a point exactly on a face of the box (orient3d == 0 counts as inside, like `<= 0.0` of pointBehindTri)
    V2[n + 3], V2[n + 4] = [2.0, 2.0, 2.0], [2.5, 2.0, 2.0]
    V2[n] = [0.5, 0.25, 0.0]
    m.set_V(V2)
    c.set_positions(V2)
    for exact in (False, True):
        m.set_exact_predicates(exact)
        c.set_exact_predicates(exact)
        assert orc.is_intersected(m) and c.is_intersected()
    c.close()
