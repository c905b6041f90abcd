"""Lagged friction through the C ABI against the oracle: multipliers / closest points / tangent bases,
friction energy, gradient and Hessian over all four stencil kinds, and the stepper with its friction lagging iterations."""
import numpy as np
import pytest

from ipc_amd import scene

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def two_blocks(gap, n=3, shift=0.13):
    Va, Fa = scene.make_box(n, 1, n, size=(1.0, 0.3, 1.0), origin=(0, 0, 0))
    Vb, Fb = scene.make_box(n, 1, n, size=(1.0, 0.3, 1.0), origin=(shift, 0.3 + gap, 0.5 * shift))
    return np.vstack([Va, Vb]), np.vstack([Fa, Fb + Va.shape[0]])


def test_friction_building_blocks(orc, gpu_lib):
    V, F = two_blocks(0.004)
    V = scene.jitter(V, F, rel=3e-3)
    SF = scene.surface_tris(F)
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_surface(SF)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    c.opt_init(0.025, False)
    c.set_surface(SF)
    dHat = 1e-3 ** 2 * m.features()["bboxDiag2"] * 160
    cs = orc.Contacts()
    sets = cs.build(m, dHat)
    act = [tuple(a) for a in sets["active"]]
    pts = [a for a in act if a[0] < 0 and a[3] >= 0]
    act.append((pts[0][0], pts[0][1], pts[0][2], -1))  # a PE and a (doubled) PP stencil so that all four kinds run
    act.append((pts[-1][0], pts[-1][1], -1, -2))
    act = np.array(act, dtype=np.int32)
    c.contact_set(act)
    fr = orc.Friction()
    lo = fr.update(m, act, 1.0, 2.0e3)
    lg = c.friction_update(1.0, 2.0e3)
    assert relerr(lg["lam"], lo["lam"]) < 1e-12 and np.abs(lg["coord"] - lo["coord"]).max() < 1e-12 and np.abs(lg["basis"] - lo["basis"]).max() < 1e-12
    rng = np.random.default_rng(5)
    U = 2e-4 * rng.normal(size=V.shape)
    U[::3] *= 30.0  # sliding on both sides of eps
    eps2, mu = (1.5e-3) ** 2, 0.37
    Vn = V + U
    m.set_V(Vn)
    c.set_positions(Vn)
    Eo = fr.energy(m, V, eps2, mu)
    assert abs(c.friction_energy(V, eps2, mu) - Eo) <= 1e-11 * abs(Eo)
    assert relerr(c.friction_gradient_add(V, eps2, mu), fr.gradient(m, V, eps2, mu)) < 1e-10
    extra = cs.connectivity(m).tolist() + [(min(-act[-2][0] - 1, act[-2][k]), max(-act[-2][0] - 1, act[-2][k])) for k in (1, 2)] \
        + [(min(-act[-1][0] - 1, act[-1][1]), max(-act[-1][0] - 1, act[-1][1]))]
    extra = np.array(extra, dtype=np.int32)
    c.set_pattern(extra)
    ia, ja = m.pattern(extra_edges=extra)
    for proj in (False, True):
        if proj:
            dbc = np.unique(np.abs(act[:6, 1]))
            m.set_dbc(dbc, 1)
            c.set_dbc(dbc, 1)
        c.set_zero()
        c.friction_hessian_add(V, eps2, mu, proj)
        a_o = fr.hessian(m, V, len(ja), eps2, mu, proj)
        assert relerr(c.get_a(), a_o) < 1e-9 and np.count_nonzero(a_o) > 0
    c.close()


def side_by_side(orc, o, c, steps, tol=1e-8, max_sub=6):
    subs = 0
    for step in range(steps):
        o.begin_timestep()
        c.begin_timestep()
        fo, fg = orc.opt_friction_state(o), c.friction_state()
        assert fg["n_lagged"] == fo["n_lagged"] and fg["n_half_space_lagged"] == fo["n_half_space_lagged"]
        assert fg["fricDHat"] == pytest.approx(fo["fricDHat"], rel=1e-14)
        if fo["n_lagged"]:
            assert relerr(fg["lam"], fo["lam"]) < 1e-9
        for sub in range(max_sub):
            for it in range(80):
                co, cg = o.newton_iter(), c.newton_iter()
                assert co == cg, (step, sub, it)
                if co:
                    break
                so, sg = o.state(), c.state()
                assert abs(sg["stepSize"] - so["stepSize"]) <= tol * so["stepSize"], (step, sub, it)
                assert abs(sg["E"] - so["E"]) <= tol * abs(so["E"]), (step, sub, it)
                assert relerr(sg["V"], so["V"]) < tol, (step, sub, it)
            else:
                pytest.fail("Newton did not converge")
            mo, mg = orc.opt_next_subproblem(o), c.next_subproblem()
            assert mo == mg, (step, sub)
            if not mo:
                break
            subs += 1
        o.end_timestep()
        c.end_timestep()
    return subs


def test_block_sliding_on_rough_ground_tracks_the_oracle(orc, gpu_lib):
    V, F = scene.make_box(2, 2, 2, size=(0.4, 0.4, 0.4), origin=(0, 0, 0))
    Vs = scene.jitter(V, F, rel=5e-3)
    SF = scene.surface_tris(F)
    vel = np.zeros_like(V)
    vel[:, 0] = 1.0
    ground = ([0, -0.004, 0], [0, 1, 0])
    m = orc.Mesh(V, F, YM=1e6, PR=0.3, density=1000.0)
    m.set_surface(SF)
    m.set_V(Vs)
    o = orc.Optimizer(m, dt=0.005, gravity=True, nthreads=2)
    orc.opt_set_half_space_friction(o, orc.opt_add_half_space(o, *ground, 5e-3), 0.5)
    orc.opt_set_velocity(o, vel)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=1e6, PR=0.3, density=1000.0)
    c.set_positions(Vs)
    c.opt_init(0.005, True)
    c.set_surface(SF)
    c.set_half_space_friction(c.add_half_space(*ground, 5e-3), 0.5)
    c.set_velocity(vel)
    o.precompute()
    c.precompute()
    side_by_side(orc, o, c, steps=16)
    assert c.friction_state()["n_half_space_lagged"] > 0
    x_mean = c.state()["V"][:, 0].mean()
    assert x_mean < Vs[:, 0].mean() + 16 * 0.005 * 1.0  # friction has slowed the block down
    c.close()


@pytest.mark.parametrize("fric_iter_amt", [1, 3])
def test_self_friction_with_lagging_iterations_tracks_the_oracle(orc, gpu_lib, fric_iter_amt):
    """Stiff block thrown sideways onto a soft mat lying on rough ground: self friction (all lagged stencil kinds that
    occur), half-space friction, and `fricIterAmt` > 1 sub-problems per time step."""
    Vm, Fm = scene.make_mat(10, thickness_ratio=0.04)
    Vb, Fb = scene.make_box(2, 2, 2, size=(0.2, 0.2, 0.2), origin=(-0.13, 0.02 + 0.012, -0.07))
    V = np.vstack([Vm, Vb])
    F = np.vstack([Fm, Fb + Vm.shape[0]]).astype(np.int32)
    nM, tM = Vm.shape[0], Fm.shape[0]
    Vs = scene.jitter(V, F, rel=5e-3)
    SF = scene.surface_tris(F)
    vel = np.zeros_like(V)
    vel[nM:, 1] = -1.0
    vel[nM:, 0] = 0.8
    ground = ([0.0, -0.024, 0.0], [0.0, 1.0, 0.0])

    m = orc.Mesh(V, F, YM=1e6, PR=0.4, density=1000.0)
    m.set_surface(SF)
    m.set_component_material((nM, V.shape[0]), (tM, F.shape[0]), 2000.0, 1e8, 0.4)
    m.set_V(Vs)
    o = orc.Optimizer(m, dt=0.01, gravity=True, nthreads=4)
    orc.opt_enable_self_collision(o, 1e-2)
    orc.opt_set_half_space_friction(o, orc.opt_add_half_space(o, *ground, 1e-2), 0.4)
    orc.opt_set_friction(o, 0.3, fric_iter_amt, 1e-3)
    orc.opt_set_velocity(o, vel)

    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=1e6, PR=0.4, density=1000.0)
    c.set_component_material((nM, V.shape[0]), (tM, F.shape[0]), 2000.0, 1e8, 0.4)
    c.set_positions(Vs)
    c.opt_init(0.01, True)
    c.set_surface(SF)
    c.enable_self_collision(1e-2)
    c.set_half_space_friction(c.add_half_space(*ground, 1e-2), 0.4)
    c.set_friction(0.3, fric_iter_amt, 1e-3)
    c.set_velocity(vel)

    o.precompute()
    c.precompute()
    subs = side_by_side(orc, o, c, steps=6)
    fs = c.friction_state()
    assert fs["n_lagged"] > 10 and fs["n_half_space_lagged"] > 10
    assert (subs > 0) == (fric_iter_amt > 1)
    c.close()
