"""Oracle parity at the sizes the bench numbers are quoted on.

* mat150 (BASELINE configs[1], 45 000 nodes / 133 206 tets): energy / gradient 1e-10, every CSR value 1e-9 with identical zero
  structure, inversion bound, one factorisation + solve against the oracle's own Cholesky, Newton iterates one by one from the
  twisted + jittered state.  At this size the patch plan has hundreds of Morton patches with halos, element slots beyond 16 bits
  of range inside a patch list, contribution lists longer than one chunk, and the solver runs its big-front path.
* 2 x mat100 stack (the scene of tools/bench_contact.py, 40 000 nodes, ~78 K active constraints): constraint sets bit-exact,
  barrier energy / gradient / PSD-projected Hessian, connectivity, CCD bounds.

Reference lines matched: Optimizer.cpp:3409-3720 (computeGradient / computePrecondMtr), SelfCollisionHandler.cpp:2149-2478.
"""
import numpy as np
import pytest

from ipc_amd import scene

pytestmark = pytest.mark.gpu

DT = 0.04
YM, PR, RHO = 2e4, 0.4, 1000.0


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.fixture(scope="module")
def mat150(orc, gpu_lib):
    V, F = scene.make_mat(150)
    Vt = scene.twist_state(scene.jitter(V, F), 0.5)
    left, right = scene.border_verts(V, 0.01)
    dbc = np.concatenate([left, right])
    m = orc.Mesh(V, F, YM=YM, PR=PR, density=RHO)
    m.set_dbc(dbc, 2)
    m.set_V(Vt)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=YM, PR=PR, density=RHO)
    c.opt_init(DT, False)
    c.set_dbc(dbc, 2)
    c.set_positions(Vt)
    yield dict(V=V, F=F, Vt=Vt, m=m, c=c, dbc=dbc, left=left, right=right)
    c.close()


def test_mat150_features_energy_gradient(mat150):
    m, c = mat150["m"], mat150["c"]
    fo, fg = m.features(), c.features()
    for k in ("restTriInv", "triArea", "mass", "mu", "lam"):
        assert relerr(fg[k], fo[k]) < 1e-13, k
    for coef in (1.0, DT * DT):
        Eo, Eg = m.elastic_energy(coef), c.elastic_energy(coef)
        assert abs(Eg - Eo) <= 1e-10 * abs(Eo)
    for proj in (True, False):
        go, gg = m.elastic_gradient(DT * DT, projectDBC=proj), c.elastic_gradient(DT * DT, projectDBC=proj)
        assert relerr(gg, go) < 1e-10
    gg = c.elastic_gradient(DT * DT, projectDBC=True).reshape(-1, 3)
    assert np.all(gg[mat150["dbc"]] == 0.0)


def test_mat150_csr_pattern_and_values(mat150):
    m, c = mat150["m"], mat150["c"]
    ia_o, ja_o = m.pattern()
    c.set_pattern()
    ia_g, ja_g = c.get_pattern()
    assert np.array_equal(ia_g, ia_o) and np.array_equal(ja_g, ja_o)
    assert len(ja_o) == 2278827
    xt = mat150["Vt"] + 1e-4 * np.random.default_rng(5).normal(size=mat150["Vt"].shape)
    c.set_xtilde(xt)
    f = m.features()
    for proj in (True, False):
        a_o = m.assemble_hessian(len(ja_o), DT * DT, projectDBC=proj)
        g_g = c.assemble_newton(DT * DT, projectDBC=proj, with_gradient=True)
        a_g = c.get_a()
        assert relerr(a_g, a_o) < 1e-9
        assert np.array_equal(a_g == 0.0, a_o == 0.0)  # same structural zeros / identity rows
        # entry-wise, not only against the largest value: every 3 x 3 block relative to its own row's diagonal scale
        rows = np.repeat(np.arange(len(ia_o) - 1), np.diff(ia_o))
        scale = np.maximum(np.abs(a_o[ia_o[:-1]])[rows], np.abs(a_o[ia_o[:-1]])[ja_o])
        assert (np.abs(a_g - a_o) / scale).max() < 1e-9
        g_o = m.elastic_gradient(DT * DT, projectDBC=proj)
        dtyp = np.zeros(mat150["V"].shape[0], dtype=int)
        dtyp[mat150["dbc"]] = 2
        free = ~((dtyp == 1) | ((dtyp == 2) & proj))
        g_o = g_o + np.repeat(f["mass"] * free, 3) * (mat150["Vt"] - xt).reshape(-1)
        assert relerr(g_g, g_o) < 1e-10
    Eo = m.elastic_energy(DT * DT) + 0.5 * (f["mass"] * ((mat150["Vt"] - xt) ** 2).sum(1)).sum()
    assert abs(c.incremental_potential(DT * DT) - Eo) <= 1e-10 * abs(Eo)


def test_mat150_inversion_bound(mat150):
    m, c = mat150["m"], mat150["c"]
    rng = np.random.default_rng(17)
    h = 1.0 / 149
    for scale in (0.05 * h, 0.5 * h, 5.0 * h):
        p = scale * rng.normal(size=3 * mat150["V"].shape[0])
        so, sg = m.filter_step_size(p, 1.0), c.filter_step_size(p, 1.0)
        assert abs(sg - so) <= 1e-9 * so
    assert c.check_inversion() and m.check_inversion()


def test_mat150_factor_solve_against_the_oracle_cholesky(mat150, orc):
    m, c = mat150["m"], mat150["c"]
    ia, ja = m.pattern()
    c.set_pattern()
    c.set_xtilde(mat150["Vt"])
    c.assemble_newton(DT * DT, True, with_gradient=False)
    a = c.get_a()
    c.analyze_pattern()
    assert c.factorize()
    b = np.random.default_rng(14).normal(size=len(ia) - 1)
    x = c.solve(b)
    assert np.linalg.norm(m.symv(a, x) - b) <= 1e-10 * np.linalg.norm(b)
    ch = orc.Chol(ia, ja, 16)
    assert ch.factorize(a)
    assert relerr(x, ch.solve(b)) < 1e-9
    # a second right-hand side through the same factor; then a not-PD matrix must be reported (CHOLMODSolver.cpp:123-154)
    b2 = np.random.default_rng(15).normal(size=len(ia) - 1)
    assert relerr(c.solve(b2), ch.solve(b2)) < 1e-9
    k = ia[3 * 20000]
    c.set_coeff(3 * 20000, 3 * 20000, -abs(a[k]))
    assert not c.factorize()
    c.set_coeff(3 * 20000, 3 * 20000, a[k])
    assert c.factorize()


def test_mat150_newton_iterates_track_the_oracle(orc, gpu_lib, mat150):
    """The state bench.py times: twist DBC on both handle columns, iterations from a twisted + jittered start."""
    V, F, Vt = mat150["V"], mat150["F"], mat150["Vt"]
    m = orc.Mesh(V, F, YM=YM, PR=PR, density=RHO)
    m.set_V(Vt)
    o = orc.Optimizer(m, dt=DT, gravity=False, nthreads=16)  # the oracle's own Cholesky scales to ~16 threads; more only costs
    o.set_twist(mat150["left"], mat150["right"], 0.4 * np.pi)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=YM, PR=PR, density=RHO)
    c.set_positions(Vt)
    c.opt_init(DT, False)
    c.set_twist(mat150["left"], mat150["right"], 0.4 * np.pi)
    o.precompute()
    c.precompute()
    o.begin_timestep()
    c.begin_timestep()
    done = 0
    for it in range(3):
        co, cg = o.newton_iter(), c.newton_iter()
        assert bool(co) == bool(cg), it
        so, sg = o.state(), c.state()
        assert relerr(sg["gradient"], so["gradient"]) < 1e-8, it
        if co:
            break
        assert abs(sg["E"] - so["E"]) <= 1e-9 * abs(so["E"]), it
        assert abs(sg["stepSize"] - so["stepSize"]) <= 1e-9 * so["stepSize"], it
        assert relerr(sg["searchDir"], so["searchDir"]) < 1e-6, it
        assert relerr(sg["V"], so["V"]) < 1e-9, it
        done += 1
    assert done >= 3
    c.close()


# ---- the contact benchmark scene at full size -----------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def stack100(orc, gpu_lib):
    V, F, nA = scene.make_mat_stack(100, 2, gap=1.2e-3)
    Vs = scene.jitter(V, F, rel=2e-3)
    SF = scene.surface_tris(F)
    border = np.nonzero((np.abs(V[:nA, 0]) > 0.49) | (np.abs(V[:nA, 2]) > 0.49))[0].astype(np.int32)
    m = orc.Mesh(V, F, YM=YM, PR=PR, density=RHO)
    m.set_surface(SF)
    m.set_dbc(border, 1)
    m.set_V(Vs)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=YM, PR=PR, density=RHO)
    c.set_dbc(border, 1)
    c.set_positions(Vs)
    c.opt_init(0.01, True)
    c.set_surface(SF)
    dHat = 1e-6 * m.features()["bboxDiag2"]
    yield dict(V=V, F=F, Vs=Vs, SF=SF, nA=nA, m=m, c=c, dHat=dHat, border=border)
    c.close()


def test_stack100_constraint_sets_bit_exact(orc, stack100):
    m, c, dHat = stack100["m"], stack100["c"], stack100["dHat"]
    svi_o, sfe_o = orc.mesh_surface(m)
    svi_g, sfe_g = c.get_surface()
    assert np.array_equal(svi_g, svi_o) and np.array_equal(sfe_g, sfe_o)
    for mult, least in ((1.0, 70000), (2.0, 100000), (0.7, 1000)):
        o = orc.Contacts().build(m, mult * dHat)
        g = c.contact_build(mult * dHat)
        assert len(o["active"]) >= least
        for k in ("active", "para", "para_eiej", "cs_ptee"):
            assert np.array_equal(g[k], o[k]), (mult, k)  # same tuples, same order, PP / PE multiplicities merged alike
    assert len(orc.Contacts().build(m, 2.0 * dHat)["para"]) > 1000  # the mollified set is exercised at this size


def test_stack100_barrier_terms(orc, stack100):
    m, c, dHat = stack100["m"], stack100["c"], stack100["dHat"]
    cs = orc.Contacts()
    sets = cs.build(m, dHat)
    c.contact_build(dHat)
    assert len(sets["active"]) > 70000
    kappa = 3.0e3
    Eo = cs.energy(m, dHat, kappa)
    assert abs(c.contact_energy(dHat, kappa) - Eo) <= 1e-10 * abs(Eo)
    for proj in (True, False):
        assert relerr(c.contact_gradient_add(dHat, kappa, proj), cs.gradient(m, dHat, kappa, proj)) < 1e-10
    pairs = c.contact_connectivity()
    assert np.array_equal(pairs, cs.connectivity(m))
    c.set_pattern(pairs)
    m2 = orc.Mesh(stack100["V"], stack100["F"], YM=YM, PR=PR, density=RHO)
    m2.set_surface(stack100["SF"])
    m2.set_dbc(stack100["border"], 1)
    m2.set_V(stack100["Vs"])
    ia, ja = m2.pattern(extra_edges=pairs)
    ia_g, ja_g = c.get_pattern()
    assert np.array_equal(ia_g, ia) and np.array_equal(ja_g, ja)
    c.set_zero()
    c.contact_hessian_add(dHat, kappa, True)
    a_o = cs.hessian(m2, len(ja), dHat, kappa, True)
    a_g = c.get_a()
    assert relerr(a_g, a_o) < 1e-9
    assert np.array_equal(a_g == 0, a_o == 0)
    # elastic + barrier: SPD, solves to the residual bound at this size
    c.assemble_newton(1e-4, True, with_gradient=False)
    c.contact_hessian_add(dHat, kappa, True)
    c.analyze_pattern()
    assert c.factorize()
    b = np.random.default_rng(2).normal(size=len(ia) - 1)
    x = c.solve(b)
    assert np.linalg.norm(c.multiply(x) - b) <= 1e-9 * np.linalg.norm(b)
    c.set_pattern()


def test_stack100_ccd_bounds(orc, stack100):
    m, c, dHat, V, nA = stack100["m"], stack100["c"], stack100["dHat"], stack100["Vs"], stack100["nA"]
    cs = orc.Contacts()
    cs.build(m, dHat)
    cand = cs.get()["cs_ptee"]
    c.contact_build(dHat)
    rng = np.random.default_rng(21)
    p = 2e-4 * rng.normal(size=V.shape)
    p[nA:, 1] -= 4e-3  # upper sheet pushed through the gap
    so, arg = orc.ccd_partial(cs, m, p.reshape(-1), 0.8, 1.0)
    sg, pg = c.ccd_partial(p.reshape(-1), 0.8, 1.0)
    assert abs(sg - so) <= 1e-12 * so and so < 1.0
    assert pg == tuple(int(x) for x in cand[arg])
    fo, pfo, no = orc.ccd_full(m, p.reshape(-1), 0.8, 1.0)
    fg, pfg, ng = c.ccd_full(p.reshape(-1), 0.8, 1.0)
    assert abs(fg - fo) <= 1e-12 * fo and pfg == pfo and ng == no
    c.set_positions(V + fg * p)
    assert not c.is_intersected()
    c.set_positions(V + p)
    m.set_V(V + p)
    assert c.is_intersected() and orc.is_intersected(m)
    c.set_positions(V)
    m.set_V(V)


def test_stack100_one_contact_newton_iteration(orc, gpu_lib, stack100):
    """One pass of solveSub_IP on the benchmark scene itself: kappa initialisation, sets, barrier terms, factor + solve,
    CCD-filtered line search."""
    V, F, Vs, SF, nA, border = (stack100[k] for k in ("V", "F", "Vs", "SF", "nA", "border"))
    vel = np.zeros_like(V)
    vel[nA:, 1] = -0.05
    m = orc.Mesh(V, F, YM=YM, PR=PR, density=RHO)
    m.set_surface(SF)
    m.set_dbc(border, 1)
    m.set_V(Vs)
    o = orc.Optimizer(m, dt=0.01, gravity=True, nthreads=16)
    orc.opt_enable_self_collision(o, 1e-3)
    orc.opt_set_velocity(o, vel)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=YM, PR=PR, density=RHO)
    c.set_dbc(border, 1)
    c.set_positions(Vs)
    c.opt_init(0.01, True)
    c.set_surface(SF)
    c.enable_self_collision(1e-3)
    c.set_velocity(vel)
    o.precompute()
    c.precompute()
    o.begin_timestep()
    c.begin_timestep()
    so, sg = o.state(), c.state()
    assert sg["dHat"] == so["dHat"]
    assert abs(sg["kappa"] - so["kappa"]) <= 1e-9 * so["kappa"]
    co, cg = o.newton_iter(), c.newton_iter()
    assert bool(co) == bool(cg)
    so, sg = o.state(), c.state()
    cst_o, cst_g = orc.opt_contact_state(o), c.contact_state()
    assert cst_g["nActive"] == len(cst_o["active"]) and cst_g["nPara"] == len(cst_o["para"])
    assert cst_g["nActive"] > 20000
    assert abs(sg["alphaFeasible"] - so["alphaFeasible"]) <= 1e-8 * so["alphaFeasible"]
    assert abs(sg["stepSize"] - so["stepSize"]) <= 1e-8 * so["stepSize"]
    assert abs(sg["kappa"] - so["kappa"]) <= 1e-8 * so["kappa"]
    assert abs(sg["E"] - so["E"]) <= 1e-8 * abs(so["E"])
    assert relerr(sg["V"], so["V"]) < 1e-8
    c.close()


# ---- contact_large: BASELINE configs[4] scale ("~1M tets: full pipeline"), SURVEY 8d item 5's stand-in --------------------------------------
# Three mat250 sheets stacked with gaps < sqrt(dHat): 375 000 nodes / 1 116 018 tets / 749 988 surface triangles, 1.36 M active constraints + 0.25 M mollified
# pairs out of 1.99 M candidate pairs at the first step.  The oracle's pieces (sets, barrier energy / gradient / Hessian, elastic energy) run at this size in seconds;
# its time stepper does not (its symbolic analysis alone takes minutes on this pattern), so the Newton iterate is held to size-independent properties
# evaluated WITH the oracle's pieces: the step solves the assembled system, the energy the stepper reports is the oracle's energy at the new state and has
# not gone up, the new state is intersection-free, and the constraint set the stepper left behind is the oracle's set at the new state.
@pytest.fixture(scope="module")
def stack250(orc, gpu_lib):
    V, F, nA = scene.make_mat_stack(250, 3, gap=1.2e-3)
    Vs = scene.jitter(V, F, rel=2e-3)
    SF = scene.surface_tris(F)
    border = np.nonzero((np.abs(V[:nA, 0]) > 0.49) | (np.abs(V[:nA, 2]) > 0.49))[0].astype(np.int32)
    m = orc.Mesh(V, F, YM=YM, PR=PR, density=RHO)
    m.set_surface(SF)
    m.set_dbc(border, 1)
    m.set_V(Vs)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=YM, PR=PR, density=RHO)
    c.set_dbc(border, 1)
    c.set_positions(Vs)
    c.opt_init(0.01, True)
    c.set_surface(SF)
    dHat = 1e-6 * m.features()["bboxDiag2"]
    yield dict(V=V, F=F, Vs=Vs, SF=SF, nA=nA, m=m, c=c, dHat=dHat, border=border)
    c.close()


def test_contact_large_constraint_sets_bit_exact(orc, stack250):
    m, c, dHat = stack250["m"], stack250["c"], stack250["dHat"]
    assert stack250["F"].shape[0] > 1100000
    o = orc.Contacts().build(m, dHat)
    g = c.contact_build(dHat)
    assert len(o["active"]) > 1300000 and len(o["para"]) > 200000 and len(o["cs_ptee"]) > 1900000  # (three 21-bit counters once capped a set at 2 M candidates)
    for k in ("active", "para", "para_eiej", "cs_ptee"):
        assert np.array_equal(g[k], o[k]), k  # same tuples, same order, PP / PE multiplicities merged alike


def test_contact_large_barrier_terms(orc, stack250):
    m, c, dHat = stack250["m"], stack250["c"], stack250["dHat"]
    cs = orc.Contacts()
    cs.build(m, dHat)
    c.contact_build(dHat)
    kappa = 3.0e3
    Eo = cs.energy(m, dHat, kappa)
    assert abs(c.contact_energy(dHat, kappa) - Eo) <= 1e-10 * abs(Eo)
    assert relerr(c.contact_gradient_add(dHat, kappa, True), cs.gradient(m, dHat, kappa, True)) < 1e-10
    pairs = c.contact_connectivity()
    assert np.array_equal(pairs, cs.connectivity(m))
    c.set_pattern(pairs)
    ia, ja = m.pattern(extra_edges=pairs)
    ia_g, ja_g = c.get_pattern()
    assert np.array_equal(ia_g, ia) and np.array_equal(ja_g, ja)
    c.set_zero()
    c.contact_hessian_add(dHat, kappa, True)
    a_g = c.get_a()
    a_o = cs.hessian(m, len(ja), dHat, kappa, True)
    assert relerr(a_g, a_o) < 1e-9
    assert np.array_equal(a_g == 0, a_o == 0)
    c.set_pattern()


def test_contact_large_newton_iterate_properties(orc, gpu_lib, stack250):
    V, F, Vs, SF, nA, border = (stack250[k] for k in ("V", "F", "Vs", "SF", "nA", "border"))
    dt = 0.01
    vel = np.zeros_like(V)
    vel[nA:, 1] = -0.05
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=YM, PR=PR, density=RHO)
    c.set_dbc(border, 1)
    c.set_positions(Vs)
    c.opt_init(dt, True)
    c.set_surface(SF)
    c.enable_self_collision(1e-3)
    c.set_velocity(vel)
    c.precompute()
    c.begin_timestep()
    s0 = c.state()
    m = orc.Mesh(V, F, YM=YM, PR=PR, density=RHO)
    m.set_surface(SF)
    m.set_dbc(border, 1)
    mass = m.features()["mass"]
    free = np.ones(V.shape[0], dtype=bool)
    free[border] = False
    xt = Vs + free[:, None] * (dt * vel + dt * dt * np.array([0.0, -9.80665, 0.0]))  # Optimizer.cpp:1236-1257

    def oracle_energy(X, kappa, dHat):
        m.set_V(X)
        cs = orc.Contacts()
        sets = cs.build(m, dHat)
        E = m.elastic_energy(dt * dt) + 0.5 * float((mass * free * ((X - xt) ** 2).sum(1)).sum()) + cs.energy(m, dHat, kappa)
        return E, sets
    assert abs(s0["dHat"] - 1e-6 * m.features()["bboxDiag2"]) <= 1e-15 * s0["dHat"]
    E0, sets0 = oracle_energy(s0["V"], s0["kappa"], s0["dHat"])
    assert abs(s0["E"] - E0) <= 1e-9 * abs(E0)
    assert c.contact_state()["nActive"] == len(sets0["active"]) > 1300000
    assert not c.newton_iter()
    s1 = c.state()
    # the search direction solves the system the iterate assembled (elastic + inertia + 1.6 M PSD-projected barrier blocks on the contact pattern)
    p, g = s1["searchDir"].reshape(-1), s1["gradient"].reshape(-1)
    assert np.linalg.norm(c.multiply(p) + g) <= 1e-8 * np.linalg.norm(g)
    assert 0.0 < s1["stepSize"] <= s1["alphaFeasible"] <= 1.0
    assert relerr(s1["V"], s0["V"] + s1["stepSize"] * s1["searchDir"].reshape(-1, 3)) < 1e-14
    # the energy the stepper accepted is the oracle's energy of the new state (kappa may have been doubled AFTER the line search: use the value it searched with)
    E1, sets1 = oracle_energy(s1["V"], s0["kappa"], s1["dHat"])
    assert abs(s1["E"] - E1) <= 1e-9 * abs(E1)
    assert E1 <= E0
    st = c.contact_state()
    assert st["nActive"] == len(sets1["active"]) and st["nPara"] == len(sets1["para"])
    assert not c.is_intersected()  # (the oracle's own check is a brute-force double loop, 8e11 pairs here; the device check is pinned against it on the 2 x mat100 stack above)
    assert c.check_inversion()
    c.close()


# ---- matTwist AS SHIPPED: input/paperExamples/14_matTwist.txt:15 says `selfCollisionOn` (BASELINE configs[1] strips it) -------------------------
def _twist_pair(orc, gpu_lib, status=None, positions=None):
    sys_path_tools()
    import bench_mat_twist as bt
    c, S = bt.make_context(150, positions=positions)
    m = orc.Mesh(S["V"], S["F"], YM=YM, PR=PR, density=RHO)
    m.set_surface(S["SF"])
    if positions is not None:
        m.set_V(positions(S["V"], S["F"]))
    o = orc.Optimizer(m, dt=DT, gravity=False, nthreads=16)
    o.set_twist(S["left"], S["right"], 0.4 * np.pi)
    orc.opt_enable_self_collision(o, 1e-3)
    if status is not None:
        c.load_status(status)
        orc.opt_load_status(o, status)
    o.precompute()
    c.precompute()
    return c, o


def sys_path_tools():
    import os
    import sys
    t = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    if t not in sys.path:
        sys.path.insert(0, t)


def _track(o, c, iters, tol):
    """one time step, iterate by iterate; returns the number of compared Newton iterations"""
    o.begin_timestep()
    c.begin_timestep()
    done = 0
    for it in range(iters):
        co, cg = o.newton_iter(), c.newton_iter()
        assert bool(co) == bool(cg), it
        if co:
            break
        so, sg = o.state(), c.state()
        cst_o, cst_g = orc_contact_state(o), c.contact_state()
        report = dict(it=it, nActive=(cst_g["nActive"], len(cst_o["active"])), nPara=(cst_g["nPara"], len(cst_o["para"])), nCand=(cst_g["nCand"], cst_o["n_candidates"]),
                      step=(sg["stepSize"], so["stepSize"]), alphaFeasible=(sg["alphaFeasible"], so["alphaFeasible"]), kappa=(sg["kappa"], so["kappa"]), E=(sg["E"], so["E"]),
                      dV=relerr(sg["V"], so["V"]), dP=relerr(sg["searchDir"], so["searchDir"]), dG=relerr(sg["gradient"], so["gradient"]),
                      dbc=(c.dbc_state(), orc_dbc_state(o)))
        assert cst_g["nActive"] == len(cst_o["active"]) and cst_g["nPara"] == len(cst_o["para"]) and cst_g["nCand"] == cst_o["n_candidates"], report
        assert abs(sg["stepSize"] - so["stepSize"]) <= tol * so["stepSize"], report
        assert abs(sg["kappa"] - so["kappa"]) <= tol * so["kappa"], report
        assert relerr(sg["V"], so["V"]) < tol, report
        assert abs(sg["E"] - so["E"]) <= tol * abs(so["E"]), report
        done += 1
    return done


def orc_dbc_state(o):
    from oracle import orc as _o
    return _o.opt_dbc_state(o)


def orc_contact_state(o):
    from oracle import orc as _o
    return _o.opt_contact_state(o)


def test_mat_twist_as_shipped_early_steps_track_the_oracle(orc, gpu_lib):
    """The first time steps of the scene an IPC user runs: nothing is active yet, but every iteration builds the constraint sets, bounds the step by CCD and
    checks for intersections (Optimizer.cpp:1884-2040, 2719-2744); the scripted twist itself is cut short by the swept hash's cap (completed step 0.43 at this
    size: SpatialHash.hpp:603-618 through AnimScripter.cpp:2158-2171) and finished by the augmented-Lagrangian fallback.  Started from a jittered, slightly
    pre-twisted state like every iterate-by-iterate test here: at exact rest (F = I) IglUtils::makePD2d is decided by round-off and the first search directions of
    ANY two implementations differ by 0.4 % -- with or without contact (measured again in round 6, tools/debug_twist_ip.py)."""
    c, o = _twist_pair(orc, gpu_lib, positions=lambda V, F: scene.twist_state(scene.jitter(V, F), 0.05))
    assert c.state()["dHat"] == o.state()["dHat"] > 0.0
    n = 0
    for step in range(2):
        n += _track(o, c, 12, 1e-9)
        o.end_timestep()
        c.end_timestep()
    assert n >= 5
    assert c.contact_state()["nActive"] == 0
    c.close()


def test_mat_twist_as_shipped_wrapped_state_tracks_the_oracle(orc, gpu_lib, tmp_path):
    """The same scene once the sheet has wrapped onto itself: the HIP stepper runs until its constraint set holds >= 1000 stencils (step 33 or so), writes the
    reference's status file (Optimizer.cpp:2964-3011); a fresh context and the oracle both restart from it (`restart`, Optimizer.cpp:179-248) and take the next
    step iterate by iterate: same set sizes, step sizes, kappa, energies, positions."""
    sys_path_tools()
    import bench_mat_twist as bt
    c, S = bt.make_context(150)
    c.precompute()
    taken = bt.advance_to_contact(c, 1000, 80)
    assert 20 <= taken < 80 and c.contact_state()["nActive"] >= 1000
    assert not c.is_intersected()
    status = str(tmp_path / "status_wrapped")
    c.save_status(status)
    c.close()
    c2, o2 = _twist_pair(orc, gpu_lib, status)
    assert c2.state()["timestep"] == o2.state()["timestep"] == taken
    n = _track(o2, c2, 3, 1e-7)
    assert n >= 3
    assert c2.contact_state()["nActive"] >= 1000
    c2.close()
