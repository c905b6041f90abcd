"""Either side of the Newton loop: Newmark time integration (Optimizer.cpp:582-590, 1259-1277,
3216-3224) and the `status<N>` text checkpoint (Optimizer.cpp:179-248, 2964-3011)."""
import numpy as np
import pytest

from ipc_amd import scene


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _free_block(orc, integ, nthreads=2):
    V, F = scene.make_box(4, 2, 2, size=(2.0, 0.5, 0.5))
    m = orc.Mesh(V, F, YM=2e5, PR=0.3, density=1000.0)
    o = orc.Optimizer(m, dt=0.004, gravity=False, nthreads=nthreads)
    if integ == "NM":
        orc.opt_set_time_integration(o, "NM")
    vel = np.zeros_like(V)
    vel[:, 0] = 1.5 * (V[:, 0] - V[:, 0].mean())  # stretching mode
    orc.opt_set_velocity(o, vel)
    o.set_rel_tol(1e-6)
    o.precompute()
    return V, F, m, o, vel


def _mech_energy(orc, m, o):
    k = orc.opt_kinematics(o)
    mass = m.features()["mass"]
    return m.elastic_energy(1.0) + 0.5 * (np.repeat(mass, 3) * k["velocity"] ** 2).sum()


def test_newmark_keeps_the_energy_backward_euler_loses(orc):
    """Trapezoidal Newmark (beta = 1/4, gamma = 1/2) is energy preserving up to the nonlinearity; BE damps."""
    res = {}
    for integ in ("BE", "NM"):
        V, F, m, o, vel = _free_block(orc, integ)
        E0 = _mech_energy(orc, m, o)
        for _ in range(60):
            assert o.solve_timestep(60) < 60
        res[integ] = _mech_energy(orc, m, o) / E0
    assert 0.97 < res["NM"] < 1.03
    assert res["BE"] < 0.8


def test_status_round_trip_in_the_oracle(orc, tmp_path):
    for integ in ("BE", "NM"):
        V, F, m, o, vel = _free_block(orc, integ)
        for _ in range(3):
            o.solve_timestep(60)
        path = tmp_path / f"status3_{integ}"
        orc.opt_save_status(o, path)
        txt = open(path).read().split()
        assert txt[:2] == ["timestep", "3"] and "position" in txt and "velocity" in txt and "acceleration" in txt and "dx_Elastic" in txt
        for _ in range(2):
            o.solve_timestep(60)
        ref = o.state()["V"]
        m2 = orc.Mesh(V, F, YM=2e5, PR=0.3, density=1000.0)
        o2 = orc.Optimizer(m2, dt=0.004, gravity=False, nthreads=2)
        if integ == "NM":
            orc.opt_set_time_integration(o2, "NM")
        o2.set_rel_tol(1e-6)
        orc.opt_load_status(o2, path)
        o2.precompute()
        assert o2.state()["timestep"] == 3
        for _ in range(2):
            o2.solve_timestep(60)
        assert relerr(o2.state()["V"], ref) < 1e-12


# ------------------------------------------------------------------------------------------------ GPU
def _gpu_pair(orc, gpu_lib, integ):
    V, F = scene.make_bar(10, 2, 2, size=(4.0, 0.5, 1.0))
    left, right = scene.border_verts(V, 0.01)
    Vs = scene.twist_state(scene.jitter(V, F, rel=2e-2), 0.1)
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_V(Vs)
    o = orc.Optimizer(m, dt=0.02, gravity=True, nthreads=4)
    o.set_twist(left, right)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    c.set_positions(Vs)
    c.opt_init(0.02, True)
    c.set_twist(left, right)
    if integ == "NM":
        orc.opt_set_time_integration(o, "NM", 0.3, 0.6)
        c.set_time_integration("NM", 0.3, 0.6)
    return V, F, Vs, left, right, m, o, c


def _step_both(o, c, steps, tol=1e-9):
    for step in range(steps):
        o.begin_timestep()
        c.begin_timestep()
        for it in range(60):
            co, cg = o.newton_iter(), c.newton_iter()
            assert bool(co) == cg, (step, it)
            if co:
                break
            so, sg = o.state(), c.state()
            assert abs(sg["E"] - so["E"]) <= tol * abs(so["E"]), (step, it)
            assert relerr(sg["V"], so["V"]) < tol, (step, it)
        else:
            pytest.fail("Newton did not converge")
        o.end_timestep()
        c.end_timestep()


@pytest.mark.gpu
@pytest.mark.parametrize("integ", ["BE", "NM"])
def test_time_integration_tracks_the_oracle(orc, gpu_lib, integ):
    V, F, Vs, left, right, m, o, c = _gpu_pair(orc, gpu_lib, integ)
    o.precompute()
    c.precompute()
    _step_both(o, c, 4)
    ko, kg = orc.opt_kinematics(o), c.kinematics()
    for k in ("velocity", "acceleration", "dx_Elastic"):
        assert relerr(kg[k], ko[k]) < 1e-7, k  # differences of positions that agree to 1e-9, divided by dt (dt^2)
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("integ", ["BE", "NM"])
def test_status_checkpoint_is_interchangeable(orc, gpu_lib, tmp_path, integ):
    V, F, Vs, left, right, m, o, c = _gpu_pair(orc, gpu_lib, integ)
    o.precompute()
    c.precompute()
    _step_both(o, c, 2)
    pg, po = tmp_path / "status_gpu", tmp_path / "status_orc"
    c.save_status(pg)
    orc.opt_save_status(o, po)
    # same grammar, same numbers
    tg, to = open(pg).read().split(), open(po).read().split()
    assert len(tg) == len(to)
    assert [t for t in tg if t[0].isalpha() and not t.startswith(("inf", "nan"))] == [t for t in to if t[0].isalpha() and not t.startswith(("inf", "nan"))]
    assert tg[:2] == ["timestep", "2"]
    # the uninterrupted pair goes on for two steps
    _step_both(o, c, 2)
    ref = c.state()["V"]
    # a fresh GPU context restarts from the ORACLE's file, a fresh oracle from the GPU's file
    V, F, Vs, left, right, m2, o2, c2 = _gpu_pair(orc, gpu_lib, integ)
    c2.load_status(po)
    orc.opt_load_status(o2, pg)
    assert c2.state()["timestep"] == 2 and o2.state()["timestep"] == 2
    o2.precompute()
    c2.precompute()
    _step_both(o2, c2, 2, tol=1e-8)
    assert relerr(c2.state()["V"], ref) < 1e-8
    c.close()
    c2.close()


@pytest.mark.gpu
def test_dirichlet_groups_track_the_oracle(orc, gpu_lib):
    """`DBC` entries of a shape line (Mesh::DirichletBCs, AnimScripter.cpp:58-110, 1440-1462): a fixed group, a group with linear
    and angular velocity that is released after two steps, iterate by iterate from a pre-strained state."""
    V, F = scene.make_bar(10, 2, 2, size=(4.0, 0.5, 1.0))
    SF = scene.surface_tris(F)
    Vs = scene.jitter(V, F, rel=2e-2)
    left = scene.select_dirichlet(V, SF, (0, 0, 0), (0.01, 1, 1))
    right = scene.select_dirichlet(V, SF, (0.99, 0, 0), (1, 1, 1))
    assert len(left) and len(right)
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_V(Vs)
    o = orc.Optimizer(m, dt=0.02, gravity=True, nthreads=4)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    c.set_positions(Vs)
    c.opt_init(0.02, True)
    for x in (o, c):
        add = (lambda *a, **k: orc.opt_add_dirichlet(o, *a, **k)) if x is o else c.add_dirichlet
        add(left)
        add(right, lin_vel=(0.3, 0.0, -0.1), ang_vel_deg=(120, 20, -35), t0=0.0, t1=0.03)
    o.precompute()
    c.precompute()
    _step_both(o, c, 4)
    Vn = c.state()["V"]
    assert np.abs(Vn[left] - Vs[left]).max() < 1e-14  # ZERO type: (x - c) + c - x, round-off only (as in the reference)
    assert np.abs(Vn[right] - Vs[right]).max() > 5e-3
    c.close()


# ------------------------------------------------------------------------------------------------ AL Dirichlet fallback
def _press_scene():
    """A kinematic block (every node in a `DBC` group moving down 1 m/s) 2 mm above an elastic slab with self-contact on: CCD
    cuts the scripted motion to a few percent, the augmented-Lagrangian penalty has to take the block the rest of the way."""
    Va, Fa = scene.make_box(3, 1, 3, size=(1.0, 0.3, 1.0), origin=(0, 0, 0))
    Vb, Fb = scene.make_box(1, 1, 1, size=(0.3, 0.3, 0.3), origin=(0.33, 0.3 + 0.002, 0.36))
    V = np.vstack([Va, Vb])
    F = np.vstack([Fa, Fb + Va.shape[0]]).astype(np.int32)
    Vs = scene.jitter(V, F, rel=3e-3)
    Vs[Va.shape[0]:] = V[Va.shape[0]:]
    ids = np.arange(Va.shape[0], V.shape[0], dtype=np.int32)
    bottom = np.nonzero(V[:Va.shape[0], 1] < 1e-9)[0].astype(np.int32)
    return V, F, Vs, scene.surface_tris(F), ids, bottom


def test_augmented_lagrangian_dirichlet_fallback_in_the_oracle(orc):
    V, F, Vs, SF, ids, bottom = _press_scene()
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_surface(SF)
    m.set_V(Vs)
    o = orc.Optimizer(m, dt=0.01, gravity=False, nthreads=4)
    orc.opt_enable_self_collision(o, 1e-3)
    orc.opt_add_dirichlet(o, ids, lin_vel=(0.0, -1.0, 0.0))
    orc.opt_add_dirichlet(o, bottom)
    o.precompute()
    for step in range(2):
        o.begin_timestep()
        d0 = orc.opt_dbc_state(o)
        assert d0["completed"] < 0.5 and d0["projectDBC"] and d0["rho"] == 0 and d0["n_targets"] == len(ids) + len(bottom)
        released = False
        for it in range(100):
            if o.newton_iter():
                break
            d = orc.opt_dbc_state(o)
            released |= (not d["projectDBC"]) and d["rho"] >= 1e6
        else:
            pytest.fail("no convergence")
        assert released and orc.opt_dbc_state(o)["completed"] > 1 - 1e-3
        o.end_timestep()
        assert np.abs(o.state()["V"][ids] - (V[ids] + [0, -0.01 * (step + 1), 0])).max() < 1e-5  # the scripted targets
    nA = len(V) - len(ids)
    under = np.nonzero((V[:nA, 1] > 0.3 - 1e-9) & (np.abs(V[:nA, 0] - 0.48) < 0.16) & (np.abs(V[:nA, 2] - 0.51) < 0.16))[0]
    assert len(under) and (o.state()["V"][under, 1] < V[ids, 1].min() - 0.02).all()  # the slab has been pressed down, no penetration


@pytest.mark.gpu
def test_augmented_lagrangian_dirichlet_fallback_tracks_the_oracle(orc, gpu_lib):
    V, F, Vs, SF, ids, bottom = _press_scene()
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_surface(SF)
    m.set_V(Vs)
    o = orc.Optimizer(m, dt=0.01, gravity=False, nthreads=4)
    orc.opt_enable_self_collision(o, 1e-3)
    orc.opt_add_dirichlet(o, ids, lin_vel=(0.0, -1.0, 0.0))
    orc.opt_add_dirichlet(o, bottom)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    c.set_positions(Vs)
    c.opt_init(0.01, False)
    c.set_surface(SF)
    c.enable_self_collision(1e-3)
    c.add_dirichlet(ids, lin_vel=(0.0, -1.0, 0.0))
    c.add_dirichlet(bottom)
    o.precompute()
    c.precompute()
    released = 0
    for step in range(3):
        o.begin_timestep()
        c.begin_timestep()
        do, dg = orc.opt_dbc_state(o), c.dbc_state()
        assert dg["n_targets"] == do["n_targets"] and abs(dg["completed"] - do["completed"]) < 1e-9
        for it in range(100):
            co, cg = o.newton_iter(), c.newton_iter()
            assert bool(co) == cg, (step, it)
            if co:
                break
            so, sg = o.state(), c.state()
            do, dg = orc.opt_dbc_state(o), c.dbc_state()
            assert dg["projectDBC"] == do["projectDBC"] and dg["rho"] == do["rho"], (step, it)
            assert abs(dg["completed"] - do["completed"]) < 1e-7, (step, it)
            assert abs(sg["E"] - so["E"]) <= 1e-8 * max(abs(so["E"]), 1e-6), (step, it)
            assert relerr(sg["V"], so["V"]) < 1e-8, (step, it)
            released += not dg["projectDBC"]
        else:
            pytest.fail("no convergence")
        o.end_timestep()
        c.end_timestep()
    assert released >= 3
    assert np.abs(c.state()["V"][ids] - (V[ids] + [0, -0.03, 0])).max() < 1e-5
    c.close()


# ------------------------------------------------------------------------------------------------ Neumann BCs
def test_neumann_on_every_node_is_gravity(orc):
    """-dt^2 m a . x on all nodes is the gravity term of x_tilde expanded: a free block under `NBC ... 0 -9.80665 0` follows the
    same trajectory as under gravity (Optimizer.cpp:3241-3250 vs. 1236-1257)."""
    V, F = scene.make_box(2, 1, 1, size=(1.0, 0.5, 0.5))
    Vs = scene.jitter(V, F, rel=1e-2)
    runs = []
    for mode in ("gravity", "nbc"):
        m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
        m.set_V(Vs)
        o = orc.Optimizer(m, dt=0.01, gravity=(mode == "gravity"), nthreads=2)
        if mode == "nbc":
            orc.opt_add_neumann(o, np.arange(V.shape[0]), (0.0, -9.80665, 0.0))
        o.set_rel_tol(1e-6)
        o.precompute()
        for _ in range(5):
            assert o.solve_timestep(60) < 60
        runs.append(o.state()["V"].copy())
    assert relerr(runs[1], runs[0]) < 1e-9
    assert runs[0][:, 1].mean() < Vs[:, 1].mean() - 1e-3  # it fell


@pytest.mark.gpu
def test_neumann_groups_track_the_oracle(orc, gpu_lib):
    """A bar clamped at one end (ZERO Dirichlet group), its other end pulled by a Neumann group for two steps, then released."""
    V, F = scene.make_bar(8, 2, 2, size=(4.0, 0.5, 1.0))
    SF = scene.surface_tris(F)
    Vs = scene.jitter(V, F, rel=2e-2)
    left = scene.select_dirichlet(V, SF, (0, 0, 0), (0.01, 1, 1))
    right = scene.select_dirichlet(V, SF, (0.9, 0, 0), (1, 1, 1))
    pull = np.concatenate([right, left[:2]])  # Dirichlet nodes inside a Neumann group are skipped
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_V(Vs)
    o = orc.Optimizer(m, dt=0.02, gravity=True, nthreads=4)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    c.set_positions(Vs)
    c.opt_init(0.02, True)
    orc.opt_add_dirichlet(o, left)
    c.add_dirichlet(left)
    orc.opt_add_neumann(o, pull, (30.0, 10.0, -5.0), t0=0.0, t1=0.03)
    c.add_neumann(pull, (30.0, 10.0, -5.0), t0=0.0, t1=0.03)
    o.precompute()
    c.precompute()
    _step_both(o, c, 4)
    assert (c.state()["V"][right, 0] > Vs[right, 0] + 1e-3).all()
    c.close()
