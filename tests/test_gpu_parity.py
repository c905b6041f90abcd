"""Parity of the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances follow BASELINE.json's north_star: integer / index outputs bit-exact, energies and
gradients 1e-10 relative, Hessian entries ~1e-9 relative (eigen-projection).
"""
import numpy as np
import pytest

from ipc_amd import scene

pytestmark = pytest.mark.gpu

REL_EG = 1e-10
REL_H = 1e-9


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def make_pair(orc, gpu_lib, V, F, Vcur=None, YM=1e5, PR=0.4, rho=1000.0, dbc=None, dt=0.025, solver=0):
    m = orc.Mesh(V, F, YM=YM, PR=PR, density=rho)
    c = gpu_lib.Context(0, solver=solver)
    c.set_mesh(V, F, YM=YM, PR=PR, density=rho)
    c.opt_init(dt=dt, gravity=False)
    if dbc is not None:
        m.set_dbc(dbc, 2)
        c.set_dbc(dbc, 2)
    if Vcur is not None:
        m.set_V(Vcur)
        c.set_positions(Vcur)
    return m, c


@pytest.fixture(scope="module")
def bar(orc, gpu_lib):
    V, F = scene.make_bar(16, 3, 3, size=(6.0, 0.75, 1.0))
    Vt = scene.twist_state(scene.jitter(V, F), 0.25)
    left, right = scene.border_verts(V, 0.01)
    dbc = np.concatenate([left, right])
    m, c = make_pair(orc, gpu_lib, V, F, Vt, dbc=dbc)
    return dict(V=V, F=F, Vt=Vt, m=m, c=c, dbc=dbc, left=left, right=right)


def test_mesh_features(bar):
    fo, fg = bar["m"].features(), bar["c"].features()
    for k in ("restTriInv", "triArea", "mass", "mu", "lam"):
        assert relerr(fg[k], fo[k]) < 1e-13, k


def test_elastic_energy(bar):
    for coef in (1.0, 0.025 ** 2):
        Eo, Eg = bar["m"].elastic_energy(coef), bar["c"].elastic_energy(coef)
        assert abs(Eg - Eo) <= REL_EG * abs(Eo)
    _, pe = bar["m"].elastic_energy(1.0, per_elem=True)
    assert relerr(bar["c"].elastic_energy_per_elem(), pe) < 1e-9  # per element: cancellation near psi ~ 0


def test_elastic_gradient(bar):
    for proj in (True, False):
        go = bar["m"].elastic_gradient(0.025 ** 2, projectDBC=proj)
        gg = bar["c"].elastic_gradient(0.025 ** 2, projectDBC=proj)
        assert relerr(gg, go) < REL_EG
        if proj:
            for v in bar["dbc"]:
                assert np.all(gg[3 * v:3 * v + 3] == 0.0)


def test_csr_pattern_bit_exact(bar):
    ia_o, ja_o = bar["m"].pattern()
    bar["c"].set_pattern()
    ia_g, ja_g = bar["c"].get_pattern()
    assert np.array_equal(ia_g, ia_o) and np.array_equal(ja_g, ja_o)
    # contact connectivity changes the pattern (SelfCollisionHandler.cpp:330-415)
    extra = np.array([[0, bar["V"].shape[0] - 1], [5, 200], [200, 5], [17, 18]], dtype=np.int32)
    V, F = bar["V"], bar["F"]
    m2 = type(bar["m"])(V, F)
    ia2, ja2 = m2.pattern(extra_edges=extra)
    bar["c"].set_pattern(extra)
    ia3, ja3 = bar["c"].get_pattern()
    assert np.array_equal(ia3, ia2) and np.array_equal(ja3, ja2)
    bar["c"].set_pattern()


def test_newton_assembly_hessian_and_gradient(bar, orc):
    m, c = bar["m"], bar["c"]
    dtSq = 0.025 ** 2
    ia, ja = m.pattern()
    c.set_pattern()
    # xTilde != x so the inertia term is exercised
    xt = bar["Vt"] + 1e-3 * np.random.default_rng(11).normal(size=bar["Vt"].shape)
    c.set_xtilde(xt)
    for proj in (True, False):
        a_o = m.assemble_hessian(len(ja), dtSq, projectDBC=proj)
        g_g = c.assemble_newton(dtSq, projectDBC=proj, with_gradient=True)
        a_g = c.get_a()
        assert relerr(a_g, a_o) < REL_H
        # identical structural zeros / identity rows
        assert np.array_equal(a_g == 0.0, a_o == 0.0)
        f = m.features()
        g_o = m.elastic_gradient(dtSq, projectDBC=proj)
        dtyp = np.zeros(bar["V"].shape[0], dtype=int)
        dtyp[bar["dbc"]] = 2
        free = ~((dtyp == 1) | ((dtyp == 2) & proj))
        g_o = g_o + (np.repeat(f["mass"] * free, 3) * (bar["Vt"] - xt).reshape(-1))
        assert relerr(g_g, g_o) < REL_EG
        assert relerr(c.gradient(dtSq, projectDBC=proj), g_o) < REL_EG
    # multiply == symmetric CSR product (LinSysSolver.hpp:238-253)
    x = np.random.default_rng(12).normal(size=len(ia) - 1)
    assert relerr(c.multiply(x), m.symv(c.get_a(), x)) < 1e-12


def test_incremental_potential(bar):
    m, c = bar["m"], bar["c"]
    dtSq = 0.025 ** 2
    xt = bar["Vt"] + 1e-3 * np.random.default_rng(11).normal(size=bar["Vt"].shape)
    c.set_xtilde(xt)
    f = m.features()
    Eo = m.elastic_energy(dtSq) + 0.5 * (f["mass"] * ((bar["Vt"] - xt) ** 2).sum(1)).sum()
    assert abs(c.incremental_potential(dtSq) - Eo) <= REL_EG * abs(Eo)


def test_filter_step_size(bar):
    m, c = bar["m"], bar["c"]
    rng = np.random.default_rng(13)
    for scale in (0.05, 0.5, 5.0):
        p = scale * rng.normal(size=3 * bar["V"].shape[0])
        so, sg = m.filter_step_size(p, 1.0), c.filter_step_size(p, 1.0)
        assert abs(sg - so) <= 1e-9 * so
    assert c.filter_step_size(np.zeros(3 * bar["V"].shape[0]), 1.0) == 1.0
    assert c.check_inversion() and m.check_inversion()


@pytest.mark.parametrize("solver", [0, 1])
def test_factorize_solve(bar, orc, gpu_lib, solver):
    m = bar["m"]
    c = gpu_lib.Context(0, solver=solver)
    c.set_mesh(bar["V"], bar["F"], YM=1e5, PR=0.4, density=1000.0)
    c.opt_init(0.025, False)
    c.set_dbc(bar["dbc"], 2)
    c.set_positions(bar["Vt"])
    c.set_pattern()
    ia, ja = m.pattern()
    c.assemble_newton(0.025 ** 2, True, with_gradient=False)
    a = c.get_a()
    c.analyze_pattern()
    assert c.factorize()
    b = np.random.default_rng(14).normal(size=len(ia) - 1)
    x = c.solve(b)
    assert np.linalg.norm(m.symv(a, x) - b) <= 1e-10 * np.linalg.norm(b)
    ch = orc.Chol(ia, ja, 4)
    assert ch.factorize(a)
    assert relerr(x, ch.solve(b)) < 1e-9
    # not positive definite -> factorize() == False and the Jacobi fallback is available (Optimizer.cpp:2331-2348)
    k = ia[3 * 40]
    c.set_coeff(3 * 40, 3 * 40, -abs(a[k]))
    assert not c.factorize()
    a2 = c.get_a()
    assert a2[k] == -abs(a[k])
    d = c.precondition_diag(b)
    assert np.allclose(d, b / a2[ia[:-1]], rtol=1e-15)
    c.close()


def test_diagnostic_known_answer_through_c_abi(gpu_lib):
    # Diagnostic.cpp:367-392: 10 isolated nodes, diagonal 10, rhs 1 => x = 0.1
    ja, ptr = [], [0]
    for v in range(10):
        for r in range(3):
            ja += [3 * v + k for k in range(r, 3)]
            ptr.append(len(ja))
    ia, ja = np.array(ptr, dtype=np.int32), np.array(ja, dtype=np.int32)
    c = gpu_lib.Context(0)
    c.set_pattern_csr(ia, ja)
    c.set_zero()
    for r in range(30):
        c.add_coeff(r, r, 10.0)
    c.add_coeff(5, 3, 99.0)  # lower-triangle writes are ignored (LinSysSolver.hpp:402-410)
    c.analyze_pattern()
    assert c.factorize()
    assert np.allclose(c.solve(np.ones(30)), 0.1, rtol=0, atol=1e-15)
    c.close()


def _pair_optimizers(orc, gpu_lib, V, F, Vstart, left, right, tol=None):
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_V(Vstart)
    o = orc.Optimizer(m, dt=0.025, gravity=False, nthreads=4)
    o.set_twist(left, right)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    c.set_positions(Vstart)
    c.opt_init(0.025, False)
    c.set_twist(left, right)
    if tol is not None:
        o.set_rel_tol(tol)
        c.set_rel_tol(tol)
    o.precompute()
    c.precompute()
    return m, o, c


def test_newton_iterates_track_the_oracle(orc, gpu_lib):
    """Iterate-by-iterate parity from a generic (jittered, pre-twisted) state.  A generic state matters: the
    reference's makePD2d (IglUtils.hpp:138-177) is discontinuous where psi_i + psi_j changes sign, i.e. exactly
    at rest, so from a rest start the projected Hessian -- in the reference as well -- depends on round-off."""
    V, F = scene.make_bar(12, 2, 2, size=(5.0, 0.5, 1.0))
    left, right = scene.border_verts(V, 0.01)
    Vs = scene.twist_state(scene.jitter(V, F, rel=2e-2), 0.15)
    m, o, c = _pair_optimizers(orc, gpu_lib, V, F, Vs, left, right)
    for step in range(2):
        o.begin_timestep()
        c.begin_timestep()
        for it in range(40):
            co, cg = o.newton_iter(), c.newton_iter()
            assert bool(co) == cg, (step, it)
            so, sg = o.state(), c.state()
            assert relerr(sg["gradient"], so["gradient"]) < 1e-7
            if co:
                break
            assert abs(sg["E"] - so["E"]) <= 1e-9 * abs(so["E"])
            assert abs(sg["stepSize"] - so["stepSize"]) <= 1e-9 * so["stepSize"]
            assert relerr(sg["searchDir"], so["searchDir"]) < 1e-6
            assert relerr(sg["V"], so["V"]) < 1e-9
        o.end_timestep()
        c.end_timestep()
    assert o.state()["innerIterAmt"] == c.state()["innerIterAmt"]
    c.close()


def test_converged_time_steps_from_rest(orc, gpu_lib):
    """From the rest state both implementations must reach the same minimiser of every incremental potential
    (tight Newton tolerance), whatever path the round-off-sensitive projection sends them on."""
    V, F = scene.make_bar(12, 2, 2, size=(5.0, 0.5, 1.0))
    left, right = scene.border_verts(V, 0.01)
    m, o, c = _pair_optimizers(orc, gpu_lib, V, F, V, left, right, tol=1e-7)
    for step in range(3):
        no, ng = o.solve_timestep(60), c.solve_timestep(60)
        assert no < 60 and ng < 60
        so, sg = o.state(), c.state()
        assert relerr(sg["V"], so["V"]) < 1e-7
        assert abs(sg["E"] - so["E"]) <= 1e-8 * abs(so["E"])
    c.close()


def test_full_size_properties_mat150(gpu_lib):
    """BASELINE config[1] size (mat150: 45 000 nodes / 133 206 tets): size-independent properties."""
    V, F = scene.make_mat(150)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
    c.opt_init(0.04, False)
    assert c.elastic_energy(1.0) == 0.0 or abs(c.elastic_energy(1.0)) < 1e-12  # rest state
    assert np.abs(c.elastic_gradient(1.0, False)).max() < 1e-9
    # rigid motions leave psi and |grad| unchanged (frame indifference)
    R = scene.rot_x(0.7)
    c.set_positions(V @ R.T + np.array([0.3, -0.2, 0.1]))
    assert abs(c.elastic_energy(1.0)) < 1e-10
    Vt = scene.twist_state(scene.jitter(V, F), 0.5)
    c.set_positions(Vt)
    E1 = c.elastic_energy(1.0)
    g1 = c.elastic_gradient(1.0, False)
    c.set_positions(Vt @ R.T)
    assert abs(c.elastic_energy(1.0) - E1) <= 1e-10 * E1
    g2 = c.elastic_gradient(1.0, False).reshape(-1, 3) @ R  # rotate back
    assert relerr(g2.reshape(-1), g1) < 1e-9
    assert abs(g1.reshape(-1, 3).sum(0)).max() < 1e-9 * np.abs(g1).max() * 100  # zero net force
    # linear solve: residual at full size
    c.set_positions(Vt)
    left, right = scene.border_verts(V, 0.01)
    c.set_dbc(np.concatenate([left, right]), 2)
    c.set_pattern()
    n, nnz = c.get_dims()
    assert n == 135000 and nnz == 6 * 45000 + 9 * ((nnz - 6 * 45000) // 9)
    c.assemble_newton(0.04 ** 2, True, with_gradient=False)
    c.analyze_pattern()
    assert c.factorize()
    b = np.random.default_rng(3).normal(size=n)
    x = c.solve(b)
    assert np.linalg.norm(c.multiply(x) - b) <= 1e-9 * np.linalg.norm(b)
    c.close()


def test_entry_destinations_on_the_device_equal_the_host_function(gpu_lib):
    """Round 5: the slot of every entry of the user's matrix in the front buffer is computed by a device kernel in the solver's set-up (k_entry_dst,
    ipc_amd/csrc/mf_numeric.hip) instead of on the host.  The host function it replaced, mf_entry_destinations (mf_symbolic.cpp, reached through the
    test shim tests/mf_symbolic/shim.cpp and itself pinned on a Python restatement in tests/test_mf_symbolic.py), must give the same numbers, bit for
    bit -- on a plain mesh pattern and on a pattern with contact pairs between two sheets."""
    import ctypes as C
    from test_mf_symbolic import analyze
    from test_sharding_gloo import _shim_lib
    shim = _shim_lib()
    for contact in (False, True):
        V, F, nA = scene.make_mat_stack(24, 2, gap=1.2e-3)
        c = gpu_lib.Context(0)
        c.set_mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
        c.opt_init(0.01, False)
        if contact:
            top = np.where((np.arange(V.shape[0]) < nA) & (V[:, 1] > V[:nA, 1].mean()))[0]
            bot = np.where((np.arange(V.shape[0]) >= nA) & (V[:, 1] < V[nA:, 1].mean()))[0]
            k = min(len(top), len(bot))
            c.set_pattern(np.stack([top[:k], bot[:k]], 1).astype(np.int32))
        else:
            c.set_pattern()
        c.analyze_pattern()
        ia, ja = c.get_pattern()
        got = c.entry_destinations()
        o = analyze(shim, np.ascontiguousarray(ia, np.int32), np.ascontiguousarray(ja, np.int32), np.ascontiguousarray(V, np.float64), leaf=12)
        assert got.shape == o["aDst"].shape and np.array_equal(got, o["aDst"]), int((got != o["aDst"]).sum())
        assert len(np.unique(got)) == len(got)  # every entry has a slot of its own
        c.close()


@pytest.mark.parametrize("block", [64, 128])
def test_two_level_blocking_of_wide_fronts(gpu_lib, block):
    """Round 5: the fronts of a level whose step launches move many bytes factor their own columns in outer blocks -- a step's rank-32 update stops at the end of its
    block, one bulk update per block (k_big_bulk, ipc_amd/csrc/mf_numeric.hip) brings the columns behind it up to date, the panel that opens the next block has nothing
    left to apply.  By default no mesh of the test suite is large enough to take that path (48 MB per step launch: meshes beyond ~200 K nodes), so it is forced on here
    for every level (ipcgpu_linsys_set_tuning: bulk_min_mb 0) with small blocks: the same matrix factorised both ways must give the same solution, and the residual of the CSR product
    must be at round-off.  Not-PD detection included (the flag travels through the same step launches)."""
    V, F = scene.make_mat(60)
    Vt = scene.twist_state(scene.jitter(V, F), 0.5)
    left, right = scene.border_verts(V, 0.01)
    xs = []
    for forced in (False, True):
        if True:
            c = gpu_lib.Context(0)
            c.set_solver_tuning(0.0, block) if forced else c.set_solver_tuning(1e9, 256)
            c.set_mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
            c.opt_init(0.04, False)
            c.set_dbc(np.concatenate([left, right]), 2)
            c.set_positions(Vt)
            c.set_pattern()
            c.assemble_newton(0.04 ** 2, True, with_gradient=False)
            c.analyze_pattern()  # (the tuning is read by the set-up of the numeric phase, here)
        assert c.factorize()
        rows, _ = c.get_dims()
        b = np.random.default_rng(5).normal(size=rows)
        x = c.solve(b)
        assert np.linalg.norm(c.multiply(x) - b) <= 1e-12 * np.linalg.norm(b)
        xs.append(x)
        if forced:
            ia, _ = c.get_pattern()
            k = 3 * (rows // 6)
            c.set_coeff(k, k, -abs(c.get_a()[ia[k]]))
            assert not c.factorize()
        c.close()
    assert relerr(xs[1], xs[0]) <= 1e-11
