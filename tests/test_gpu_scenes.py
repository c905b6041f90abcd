"""Small stand-ins of BASELINE.json configs 3 and 4 run iterate by iterate against the oracle through the
C ABI: every piece of the path at once -- two materials, self-contact, a half-space, gravity (config 3: stiff block on a
soft mat over the ground) and scripted twist handles with heavy self-contact incl. the mollified parallel-edge set
(config 4: two rods twisted around each other)."""
import numpy as np
import pytest

from ipc_amd import scene

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def run_side_by_side(orc, o, c, steps, max_iter=80, tol=1e-8):
    seen = dict(active=0, para=0, half=0, full_ccd=0, limited=0)
    for step in range(steps):
        o.begin_timestep()
        c.begin_timestep()
        assert abs(c.state()["kappa"] - o.state()["kappa"]) <= tol * o.state()["kappa"], step
        for it in range(max_iter):
            co, cg = o.newton_iter(), c.newton_iter()
            assert co == cg, (step, it)
            if co:
                break
            so, sg = o.state(), c.state()
            ko, kg = orc.opt_contact_state(o), c.contact_state()
            assert kg["nActive"] == len(ko["active"]) and kg["nPara"] == len(ko["para"]), (step, it)
            assert abs(sg["alphaFeasible"] - so["alphaFeasible"]) <= tol * so["alphaFeasible"], (step, it)
            assert abs(sg["stepSize"] - so["stepSize"]) <= tol * so["stepSize"], (step, it)
            assert abs(sg["kappa"] - so["kappa"]) <= tol * so["kappa"], (step, it)
            assert abs(sg["E"] - so["E"]) <= tol * abs(so["E"]), (step, it)
            assert relerr(sg["V"], so["V"]) < tol, (step, it)
            seen["active"] = max(seen["active"], kg["nActive"])
            seen["para"] = max(seen["para"], kg["nPara"])
            seen["half"] = max(seen["half"], kg["nHalfSpace"])
            seen["limited"] += so["alphaFeasible"] < 1.0
        else:
            pytest.fail("Newton did not converge")
        o.end_timestep()
        c.end_timestep()
        assert c.check_inversion() and not c.is_intersected()
    seen["full_ccd"] = c.contact_state()["nFullCCD"]
    assert seen["full_ccd"] == orc.opt_contact_state(o)["n_full_ccd"]
    return seen


def test_config3_stiff_block_on_soft_mat_over_ground(orc, gpu_lib):
    Vm, Fm = scene.make_mat(10, thickness_ratio=0.04)
    Vb, Fb = scene.make_box(2, 2, 2, size=(0.2, 0.2, 0.2), origin=(-0.13, 0.02 + 0.012, -0.07))
    V = np.vstack([Vm, Vb])
    F = np.vstack([Fm, Fb + Vm.shape[0]]).astype(np.int32)
    nM, tM = Vm.shape[0], Fm.shape[0]
    Vs = scene.jitter(V, F, rel=5e-3)
    SF = scene.surface_tris(F)
    vel = np.zeros_like(V)
    vel[nM:, 1] = -1.0
    ground = ([0.0, -0.06, 0.0], [0.0, 1.0, 0.0])

    m = orc.Mesh(V, F, YM=1e6, PR=0.4, density=1000.0)
    m.set_surface(SF)
    m.set_component_material((nM, V.shape[0]), (tM, F.shape[0]), 2000.0, 1e8, 0.4)  # 12_sphereOnMat.txt materials
    m.set_V(Vs)
    o = orc.Optimizer(m, dt=0.01, gravity=True, nthreads=4)
    orc.opt_enable_self_collision(o, 1e-2)
    orc.opt_add_half_space(o, *ground, 1e-2)
    orc.opt_set_velocity(o, vel)

    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=1e6, PR=0.4, density=1000.0)
    c.set_component_material((nM, V.shape[0]), (tM, F.shape[0]), 2000.0, 1e8, 0.4)
    fo, fg = m.features(), c.features()
    assert np.array_equal(fg["mu"], fo["mu"]) and np.array_equal(fg["lam"], fo["lam"]) and relerr(fg["mass"], fo["mass"]) < 1e-15
    c.set_positions(Vs)
    c.opt_init(0.01, True)
    c.set_surface(SF)
    c.enable_self_collision(1e-2)
    c.add_half_space(*ground, 1e-2)
    c.set_velocity(vel)

    o.precompute()
    c.precompute()
    seen = run_side_by_side(orc, o, c, steps=8)
    assert seen["active"] > 20 and seen["half"] > 50
    Vn = c.state()["V"]
    assert Vn[:, 1].min() > -0.06 and Vn[nM:, 1].min() > Vn[:nM, 1].min()
    c.close()


def test_config4_two_rods_twisted_together(orc, gpu_lib):
    Va, Fa = scene.make_box(12, 1, 1, size=(3.0, 0.25, 0.25), origin=(-1.5, -0.125, 0.01))
    Vb, Fb = scene.make_box(12, 1, 1, size=(3.0, 0.25, 0.25), origin=(-1.5, -0.125, -0.26))
    V = np.vstack([Va, Vb])
    F = np.vstack([Fa, Fb + Va.shape[0]]).astype(np.int32)
    Vs = scene.jitter(V, F, rel=5e-3)
    SF = scene.surface_tris(F)
    left, right = scene.border_verts(V, 0.01)

    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_surface(SF)
    m.set_V(Vs)
    o = orc.Optimizer(m, dt=0.025, gravity=False, nthreads=4)
    o.set_twist(left, right, 0.4 * np.pi)
    orc.opt_enable_self_collision(o, 1e-2)

    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    c.set_positions(Vs)
    c.opt_init(0.025, False)
    c.set_surface(SF)
    c.set_twist(left, right, 0.4 * np.pi)
    c.enable_self_collision(1e-2)

    o.precompute()
    c.precompute()
    seen = run_side_by_side(orc, o, c, steps=8)
    assert seen["active"] > 50 and seen["para"] > 0
    c.close()


# ------------------------------------------------------------------------------------------------ config 0
def _config0():
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "config0_bar2523.npz"))
    return {k: g[k] for k in g.files}


def test_config0_hello_world_on_the_reference_mesh(orc, gpu_lib):
    """BASELINE configs[0]: input/otherExamples/barTwist_noCollisions.txt on the reference's own bar-2523.msh (886 nodes,
    2 523 tets, E = 1e9): left end fixed, right end turning 270 deg/s as a `DBC` group.  The mesh arrays and the expected
    positions are the committed fixture of tools/make_golden_config0.py; the live oracle runs beside it as well."""
    g = _config0()
    V, T = g["V"], g["T"]
    assert V.shape == (886, 3) and T.shape == (2523, 4)
    m = orc.Mesh(V, T, YM=1e9, PR=0.4, density=1000.0)
    o = orc.Optimizer(m, dt=0.025, gravity=True, nthreads=4)
    orc.opt_add_dirichlet(o, g["left"])
    orc.opt_add_dirichlet(o, g["right"], ang_vel_deg=(270, 0, 0))
    c = gpu_lib.Context(0)
    c.set_mesh(V, T, YM=1e9, PR=0.4, density=1000.0)
    c.opt_init(0.025, True)
    c.add_dirichlet(g["left"])
    c.add_dirichlet(g["right"], ang_vel_deg=(270, 0, 0))
    # from an exact rest start single iterates are round-off dependent (makePD2d, see tools/make_golden_config0.py): compare
    # the converged time steps at the fixture's tight tolerance
    o.set_rel_tol(float(g["rel_tol"]))
    c.set_rel_tol(float(g["rel_tol"]))
    o.precompute()
    c.precompute()
    for step in range(len(g["iters"])):
        no, ng = o.solve_timestep(100), c.solve_timestep(100)
        assert no < 100 and ng < 100
        so, sg = o.state(), c.state()
        assert relerr(sg["V"], so["V"]) < 1e-6 and relerr(sg["V"], g["positions"][step]) < 1e-6
        assert abs(sg["E"] - g["energy"][step]) <= 1e-6 * abs(g["energy"][step])
    # the right end has really turned: 3 steps of 270 deg/s * 0.025 s
    th = np.deg2rad(270 * 0.025 * len(g["iters"]))
    ctr = 0.5 * (V[g["right"]].min(0) + V[g["right"]].max(0))
    y0, z0 = (V[g["right"], 1] - ctr[1]), (V[g["right"], 2] - ctr[2])
    Vn = c.state()["V"]
    assert np.allclose(Vn[g["right"], 1] - ctr[1], np.cos(th) * y0 - np.sin(th) * z0, atol=1e-9)
    assert np.allclose(Vn[g["right"], 2] - ctr[2], np.sin(th) * y0 + np.cos(th) * z0, atol=1e-9)
    assert np.allclose(Vn[g["left"]], V[g["left"]], rtol=0, atol=1e-14)
    c.close()


@pytest.mark.parametrize("option,tit", [(1, "BE"), (2, "BE"), (3, "BE"), (4, "NM")])
def test_warm_start_options_track_the_oracle(orc, gpu_lib, option, tit):
    """Config `warmStart n` -> Optimizer::initX(n) (Optimizer.cpp:925-1215): the predicted first iterate, cut by the inversion
    filter, the half-space bound and a full CCD pass.  A block thrown at the ground and at a second block: the admitted fraction
    of the prediction and every Newton iterate after it agree with the oracle."""
    Va, Fa = scene.make_box(2, 1, 2, size=(0.6, 0.2, 0.6), origin=(-0.3, 0.02, -0.3))
    Vb, Fb = scene.make_box(1, 1, 1, size=(0.2, 0.2, 0.2), origin=(-0.07, 0.26, -0.11))
    V = np.vstack([Va, Vb])
    F = np.vstack([Fa, Fb + Va.shape[0]]).astype(np.int32)
    Vs = scene.jitter(V, F, rel=1e-2)
    SF = scene.surface_tris(F)
    vel = np.zeros_like(V)
    vel[:, 1] = -1.5
    vel[Va.shape[0]:, 1] = -4.0  # the small block catches up with the slab
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_surface(SF)
    m.set_V(Vs)
    o = orc.Optimizer(m, dt=0.01, gravity=True, nthreads=2)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    c.set_positions(Vs)
    c.opt_init(0.01, True)
    c.set_surface(SF)
    if tit == "NM":
        orc.opt_set_time_integration(o, "NM")
        c.set_time_integration("NM")
    orc.opt_enable_self_collision(o, 1e-2)
    c.enable_self_collision(1e-2)
    orc.opt_add_half_space(o, [0, 0, 0], [0, 1, 0], 1e-2)
    c.add_half_space([0, 0, 0], [0, 1, 0], 1e-2)
    orc.opt_set_velocity(o, vel)
    c.set_velocity(vel)
    orc.opt_set_warm_start(o, option)
    c.set_warm_start(option)
    o.precompute()
    c.precompute()
    cut = 0
    for step in range(5):
        o.begin_timestep()
        c.begin_timestep()
        wo, wg = orc.opt_warm_step(o), c.warm_step()
        assert abs(wg - wo) <= 1e-9 * wo and 0 < wo <= 1.0, (step, wo, wg)
        cut += wo < 1.0
        assert relerr(c.state()["V"], o.state()["V"]) < 1e-10, step  # the first iterate itself
        for it in range(60):
            co, cg = o.newton_iter(), c.newton_iter()
            assert co == cg, (step, it)
            if co:
                break
            so, sg = o.state(), c.state()
            assert abs(sg["stepSize"] - so["stepSize"]) <= 1e-8 * so["stepSize"], (step, it)
            assert abs(sg["E"] - so["E"]) <= 1e-8 * abs(so["E"]), (step, it)
            assert relerr(sg["V"], so["V"]) < 1e-8, (step, it)
        else:
            pytest.fail("Newton did not converge")
        o.end_timestep()
        c.end_timestep()
    assert cut > 0  # the prediction was cut by a step bound at least once (ground / CCD)
    c.close()
