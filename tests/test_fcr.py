"""Fixed corotated elasticity (`energy FCR`; FixedCoRotEnergy.cpp:62-153): oracle self-consistency
on the CPU, GPU parity through the C ABI."""
import numpy as np
import pytest

from ipc_amd import scene


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _block(orc, amp=0.04, seed=3):
    V, F = scene.make_box(3, 2, 2, size=(1.5, 1.0, 1.0))
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_energy_type("FCR")
    X = V + amp * np.random.default_rng(seed).normal(size=V.shape)
    m.set_V(X)
    return V, F, X, m


def test_fcr_energy_vanishes_under_rigid_motion(orc):
    V, F, X, m = _block(orc)
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    m.set_V(V @ R.T + np.array([0.3, -1.0, 2.0]))
    assert abs(m.elastic_energy()) < 1e-18 * 1e5
    assert np.abs(m.elastic_gradient(projectDBC=False)).max() < 1e-9


def test_fcr_gradient_matches_energy_fd(orc):
    V, F, X, m = _block(orc)
    g = m.elastic_gradient(projectDBC=False)
    rng = np.random.default_rng(5)
    for k in rng.integers(0, 3 * V.shape[0], 12):
        v, c = divmod(int(k), 3)
        Xp = X.copy()
        Xp[v, c] += 1e-6
        m.set_V(Xp)
        Ep = m.elastic_energy()
        Xp[v, c] -= 2e-6
        m.set_V(Xp)
        Em = m.elastic_energy()
        assert abs((Ep - Em) / 2e-6 - g[k]) <= 1e-6 * max(1.0, abs(g[k]))


def test_fcr_unprojected_element_hessian_matches_gradient_fd(orc):
    """Gradient of the whole mesh differentiated w.r.t. the 12 dofs of one element, against the sum of the element Hessians
    that touch those dofs restricted to that element's block (one element isolated by using a single-tet mesh)."""
    V = np.array([[0, 0, 0], [1.0, 0, 0], [0, 1.1, 0], [0.1, 0.2, 0.9]])
    F = np.array([[0, 1, 2, 3]], dtype=np.int32)
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_energy_type("FCR")
    X = V + 0.08 * np.random.default_rng(2).normal(size=V.shape)
    m.set_V(X)
    H = m.elastic_hessian_elem(0, 1.0, projectSPD=False)
    assert np.allclose(H, H.T, atol=1e-9 * np.abs(H).max())
    Hfd = np.zeros((12, 12))
    for k in range(12):
        v, c = divmod(k, 3)
        Xp = X.copy()
        Xp[v, c] += 1e-6
        m.set_V(Xp)
        gp = m.elastic_gradient(projectDBC=False)
        Xp[v, c] -= 2e-6
        m.set_V(Xp)
        gm = m.elastic_gradient(projectDBC=False)
        Hfd[:, k] = (gp - gm) / 2e-6
    assert np.abs(H - Hfd).max() <= 1e-5 * np.abs(H).max()
    Hp = m.elastic_hessian_elem(0, 1.0, projectSPD=True)
    assert np.linalg.eigvalsh(Hp).min() >= -1e-9 * np.abs(Hp).max()


def test_fcr_survives_an_inverted_element(orc):
    """No element-inversion safeguard for FCR (Energy<dim>(false)): energy / gradient stay finite through det F < 0 and the
    step filter leaves the step alone (Energy.cpp:565-581)."""
    V, F, X, m = _block(orc)
    Xi = X.copy()
    Xi[F[0, 3]] += 3.0 * (X[F[0, :3]].mean(0) - X[F[0, 3]])  # push a vertex through the opposite face
    m.set_V(Xi)
    assert not m.check_inversion()
    assert np.isfinite(m.elastic_energy()) and np.isfinite(m.elastic_gradient(projectDBC=False)).all()
    assert m.filter_step_size(np.ones(3 * V.shape[0]), 1.0) == 1.0


def test_fcr_newton_converges(orc):
    V, F = scene.make_bar(8, 2, 2, size=(4.0, 0.5, 1.0))
    left, right = scene.border_verts(V, 0.01)
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_energy_type("FCR")
    m.set_V(scene.twist_state(scene.jitter(V, F, rel=2e-2), 0.1))
    o = orc.Optimizer(m, dt=0.025, gravity=False, nthreads=2)
    o.set_twist(left, right)
    o.precompute()
    for _ in range(2):
        assert o.solve_timestep(60) < 60


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_fcr_building_blocks_on_the_gpu(orc, gpu_lib):
    V, F = scene.make_bar(10, 3, 3, size=(4.0, 0.75, 1.0))
    Vt = scene.twist_state(scene.jitter(V, F, rel=3e-2), 0.2)
    left, right = scene.border_verts(V, 0.01)
    dbc = np.concatenate([left, right])
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_energy_type("FCR")
    m.set_dbc(dbc, 2)
    m.set_V(Vt)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    c.set_energy_type("FCR")
    c.opt_init(dt=0.025, gravity=False)
    c.set_dbc(dbc, 2)
    c.set_positions(Vt)
    dtSq = 0.025 ** 2
    Eo, pe = m.elastic_energy(1.0, per_elem=True)
    assert abs(c.elastic_energy(1.0) - Eo) <= 1e-10 * abs(Eo)
    assert relerr(c.elastic_energy_per_elem(), pe) < 1e-9
    for proj in (True, False):
        assert relerr(c.elastic_gradient(dtSq, projectDBC=proj), m.elastic_gradient(dtSq, projectDBC=proj)) < 1e-10
    ia, ja = m.pattern()
    c.set_pattern()
    c.set_xtilde(Vt)
    for proj in (True, False):
        a_o = m.assemble_hessian(len(ja), dtSq, projectDBC=proj)
        c.assemble_newton(dtSq, projectDBC=proj, with_gradient=True)
        a_g = c.get_a()
        assert relerr(a_g, a_o) < 1e-9
        assert np.array_equal(a_g == 0.0, a_o == 0.0)
    # an inverted element: still finite and still the oracle's numbers, and no step filtering
    Vi = Vt.copy()
    Vi[F[7, 3]] += 2.5 * (Vt[F[7, :3]].mean(0) - Vt[F[7, 3]])
    m.set_V(Vi)
    c.set_positions(Vi)
    assert not m.check_inversion()
    Eo = m.elastic_energy(1.0)
    assert abs(c.elastic_energy(1.0) - Eo) <= 1e-10 * abs(Eo)
    assert relerr(c.elastic_gradient(1.0, projectDBC=False), m.elastic_gradient(1.0, projectDBC=False)) < 1e-9
    a_o = m.assemble_hessian(len(ja), dtSq, projectDBC=True)
    c.assemble_newton(dtSq, projectDBC=True, with_gradient=False)
    assert relerr(c.get_a(), a_o) < 1e-8
    p = np.random.default_rng(1).normal(size=3 * V.shape[0])
    assert c.filter_step_size(p, 1.0) == 1.0 and m.filter_step_size(p, 1.0) == 1.0
    c.close()


@pytest.mark.gpu
def test_fcr_newton_iterates_track_the_oracle(orc, gpu_lib):
    V, F = scene.make_bar(12, 2, 2, size=(5.0, 0.5, 1.0))
    left, right = scene.border_verts(V, 0.01)
    Vs = scene.twist_state(scene.jitter(V, F, rel=2e-2), 0.15)
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_energy_type("FCR")
    m.set_V(Vs)
    o = orc.Optimizer(m, dt=0.025, gravity=False, nthreads=4)
    o.set_twist(left, right)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    c.set_energy_type("FCR")
    c.set_positions(Vs)
    c.opt_init(0.025, False)
    c.set_twist(left, right)
    o.precompute()
    c.precompute()
    for step in range(3):
        o.begin_timestep()
        c.begin_timestep()
        for it in range(40):
            co, cg = o.newton_iter(), c.newton_iter()
            assert bool(co) == cg, (step, it)
            if co:
                break
            so, sg = o.state(), c.state()
            assert abs(sg["E"] - so["E"]) <= 1e-9 * abs(so["E"])
            assert abs(sg["stepSize"] - so["stepSize"]) <= 1e-9 * so["stepSize"]
            assert relerr(sg["V"], so["V"]) < 1e-9
        else:
            pytest.fail("Newton did not converge")
        o.end_timestep()
        c.end_timestep()
    assert o.state()["innerIterAmt"] == c.state()["innerIterAmt"]
    c.close()
