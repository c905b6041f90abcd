"""Scene scripts (src/Config.cpp:97-620) driving the C ABI: grammar, shape assembly, and the tutorial
scene run on the GPU beside the oracle through the same calls."""
import os

import numpy as np
import pytest

from ipc_amd import lib as gl
from ipc_amd import scene, scene_script as ss

REF_INPUT = "/root/reference/input"

TUTORIAL = """
shapes input 2
cube.msh 0 3 0  0 0 0  1 1 1
cube.msh 0 1 0  0 0 0  1 1 1

selfFric 0.1

ground 0.1 0
"""  # input/tutorialExamples/2cubesFall.txt with a local mesh path

HELLO = """
energy NH
density 1000
stiffness 1e9 0.4

shapes input 1
bar.msh 0 0 0  0 0 0  1 1 1 DBC 0 0 0  0.01 1 1  0 0 0  0 0 0  DBC 0.99 0 0  1 1 1  0 0 0  270 0 0

selfCollisionOff
"""  # input/otherExamples/barTwist_noCollisions.txt


def test_grammar():
    c = ss.SceneConfig.parse(TUTORIAL, "/x")
    assert c.energy == "NH" and c.dt == 0.025 and c.YM == 1e5 and c.self_collision and c.self_fric == 0.1
    assert len(c.shapes) == 2 and c.shapes[0].path == "/x/cube.msh" and np.allclose(c.shapes[0].translate, [0, 3, 0])
    (o, n, mu), = c.half_spaces
    assert np.allclose(o, [0, 0, 0]) and np.allclose(n, [0, 1, 0]) and mu == 0.1
    c = ss.SceneConfig.parse(HELLO)
    assert c.YM == 1e9 and not c.self_collision and len(c.shapes[0].dbc) == 2
    assert c.shapes[0].dbc[1][3] == [270.0, 0.0, 0.0] and c.shapes[0].dbc[1][0] == [0.99, 0.0, 0.0]
    c = ss.SceneConfig.parse("energy FCR\ntimeIntegration NM 0.3 0.6\ntime 2 0.01\nturnOffGravity\ntol 1\n1e-4\ndHat 2e-3\nepsv 1e-4\nfricIterAmt 3\n"
                             "halfSpace 0 -1 0  0 2 0  1 0.5\nshapes input 1\n# comment\nm.msh 1 2 3  0 0 90  2 2 2 material 2000 1e8 0.3 initVel 1 0 0  0 0 90 "
                             "linearVelocity 0 -1 0\nview orthographic\n")
    assert (c.energy, c.time_integration, c.beta, c.gamma, c.dt, c.gravity, c.tol, c.dHat_eps, c.eps_v, c.fric_iter_amt) == ("FCR", "NM", 0.3, 0.6, 0.01, False, 1e-4, 2e-3, 1e-4, 3)
    assert np.allclose(c.half_spaces[0][1], [0, 1, 0]) and c.half_spaces[0][2] == 0.5
    sh = c.shapes[0]
    assert sh.material == (2000.0, 1e8, 0.3) and sh.init_vel == ((1.0, 0.0, 0.0), (0.0, 0.0, 90.0)) and sh.lin_vel == (0.0, -1.0, 0.0)
    c = ss.SceneConfig.parse("shapes input 1\nm.msh 0 0 0 0 0 0 1 1 1 NBC 0.9 0 0 1 1 1 0 -5 0 0.1 0.3\n")
    assert c.shapes[0].nbc == [([0.9, 0.0, 0.0], [1.0, 1.0, 1.0], [0.0, -5.0, 0.0], 0.1, 0.3)]
    c = ss.SceneConfig.parse("meshCO input/triMeshes/plane.obj 0.5 0 0.5  10  50  1.0 rotate 0 0 30\n", "/r")
    (path, origin, scale, mu, rot), = c.mesh_cos
    assert path == "/r/input/triMeshes/plane.obj" and np.allclose(origin, [0.5, 0, 0.5]) and scale == 10 and mu == 1.0 and np.allclose(rot, [0, 0, 30])
    c = ss.SceneConfig.parse("shapes input 1\nm.seg 0 0 0 0 0 0 1 1 1 meshSeq dir\n", "/r")
    assert c.shapes[0].mesh_seq == "/r/dir"
    for name in ss.HOLD_SCRIPTS + ss.PULL_SCRIPTS + ss.INITVEL_SCRIPTS + ss.RULE_SCRIPTS:
        assert ss.SceneConfig.parse(f"script {name}\n").script == name
    c = ss.SceneConfig.parse("DBCTimeRange 0.1 0.5\nNBCTimeRange 0.2 1\n")
    assert c.dbc_time_range == (0.1, 0.5) and c.nbc_time_range == (0.2, 1.0)
    c = ss.SceneConfig.parse("useAbsParameters\nminBarrierStiffnessScale 2e10\ntuning 4\n0 1e-2 1e-3\n2e-10\n")
    assert c.use_abs_parameters and c.kappa_min_multiplier == 2e10 and (c.dHat_eps, c.dHat_target, c.dtol_rel, c.eps_v) == (1e-2, 1e-3, 2e-10, 1e-3)
    for bad in ("script MCOSquash\n", "constraintSolver QP\n", "CCDMethod TightInclusion\n"):
        with pytest.raises(ss.UnsupportedKeyword):
            ss.SceneConfig.parse(bad)


def test_assemble_transforms_and_selects_like_main_cpp():
    V0, F0 = scene.make_box(2, 1, 1, size=(2.0, 1.0, 1.0), origin=(0, 0, 0))
    SF0 = scene.surface_tris(F0)
    c = ss.SceneConfig.parse("shapes input 2\na.msh 0 0 0  0 0 0  1 1 1 DBC 0 0 0  0.01 1 1  0 0 0  0 0 0\n"
                             "a.msh 5 0 0  0 0 90  2 1 1 initVel 0 1 0  0 0 0 angularVelocity 0 0 0\n")
    sc = ss.assemble(c, lambda p: (V0.copy(), F0.copy(), SF0.copy()))
    n = V0.shape[0]
    assert sc.V.shape[0] == 2 * n and sc.T.max() == 2 * n - 1 and sc.node_ranges == [0, n, 2 * n] and sc.tet_ranges == [0, len(F0), 2 * len(F0)]
    # R (p * scale) + t with Rz(90): (x, y, z) -> (-y, 2x, z) + (5, 0, 0)
    assert np.allclose(sc.V[n:], np.c_[-V0[:, 1] + 5, 2 * V0[:, 0], V0[:, 2]])
    ids0 = sc.dirichlet[0][0]
    assert np.array_equal(ids0, np.nonzero(V0[:, 0] < 0.02)[0])
    assert np.array_equal(sc.dirichlet[1][0], np.arange(n, 2 * n))  # scripted component: all of its nodes
    assert not sc.velocity.any()  # initVel is not applied to Dirichlet nodes (AnimScripter.cpp:1327)


def test_dirichlet_and_neumann_boxes_follow_the_shape_through_its_transform():
    """main.cpp:1045-1068 picks the nodes on the mesh as read and transforms the shape afterwards (:1073-1077): a shape rotated
    by 90 degrees about z (and mirrored by a negative scale) keeps the same node ids in its Dirichlet / Neumann sets."""
    V0, F0 = scene.make_box(4, 1, 1, size=(4.0, 1.0, 1.0), origin=(0, 0, 0))
    SF0 = scene.surface_tris(F0)
    want_dbc = np.nonzero(V0[:, 0] < 0.05)[0]  # the x = 0 face of the untransformed bar
    want_nbc = np.nonzero(V0[:, 0] > 3.95)[0]
    for rot, scl in (("0 0 90", "1 1 1"), ("0 0 90", "-1 2 1"), ("0 90 0", "1 1 1"), ("0 0 0", "1 1 1")):
        c = ss.SceneConfig.parse(f"shapes input 1\na.msh 1 2 3  {rot}  {scl} DBC 0 0 0  0.01 1 1  0 0 0  0 0 0 NBC 0.99 0 0  1 1 1  0 -3 0\n")
        sc = ss.assemble(c, lambda p: (V0.copy(), F0.copy(), SF0.copy()))
        assert np.array_equal(np.sort(sc.dirichlet[0][0]), want_dbc), (rot, scl)
        assert np.array_equal(np.sort(sc.neumann[0][0]), want_nbc), (rot, scl)
    # with the 90-degree rotation the selected face is NOT the low-x face of the transformed shape (what a selection after the
    # transform would have picked): it is a y-extreme face
    c = ss.SceneConfig.parse("shapes input 1\na.msh 0 0 0  0 0 90  1 1 1 DBC 0 0 0  0.01 1 1  0 0 0  0 0 0\n")
    sc = ss.assemble(c, lambda p: (V0.copy(), F0.copy(), SF0.copy()))
    sel = sc.V[sc.dirichlet[0][0]]
    assert np.ptp(sel[:, 1]) < 1e-12 and np.ptp(sel[:, 0]) > 0.5


@pytest.mark.skipif(not os.path.isdir(REF_INPUT), reason="the reference's scene files are only present in the build container")
def test_reference_scene_files_parse():
    ok, unsupported = [], {}
    for root, _, files in os.walk(REF_INPUT):
        for f in files:
            if not f.endswith(".txt"):
                continue
            p = os.path.join(root, f)
            try:
                c = ss.SceneConfig.parse(open(p, errors="replace").read(), "/root/reference")
                if c.shapes:
                    ok.append(os.path.relpath(p, REF_INPUT))
            except ss.UnsupportedKeyword as e:
                unsupported[str(e).split()[0]] = unsupported.get(str(e).split()[0], 0) + 1
            except (ValueError, IndexError):
                unsupported["malformed"] = unsupported.get("malformed", 0) + 1
    print(len(ok), "scene files map onto the C ABI;", unsupported)
    for need in ("tutorialExamples/2cubesFall.txt", "otherExamples/barTwist_noCollisions.txt", "paperExamples/4_rodsTwist.txt", "paperExamples/14_matTwist.txt"):
        assert need in ok, need
    assert len(ok) >= 142  # the rest asks for another CCD method or another constraint solver


class OracleBackend:
    """The same method names as ipc_amd.lib.Context over the CPU oracle, so that one `apply()` drives both."""

    def __init__(self, orc, nthreads=4):
        self.orc, self.nthreads, self.m, self.o = orc, nthreads, None, None
        self._pending = []

    def set_mesh(self, V, T, YM, PR, density):
        self.m = self.orc.Mesh(V, T, YM=YM, PR=PR, density=density)

    def set_energy_type(self, name):
        self.m.set_energy_type(name)

    def set_positions(self, V):
        self.m.set_V(V)

    def set_component_material(self, *a):
        self.m.set_component_material(*a)

    def set_surface(self, SF, codim_edges=None):
        self.m.set_surface(SF, codim_edges)

    def opt_init(self, dt, gravity):
        self.o = self.orc.Optimizer(self.m, dt=dt, gravity=gravity, nthreads=self.nthreads)

    def set_dbc(self, ids, typ):
        self.m.set_dbc(ids, typ)

    def set_warm_start(self, option):
        self.orc.opt_set_warm_start(self.o, option)

    def set_obstacle(self, ids, obstacle_only=False):
        self.m.set_obstacle(ids, obstacle_only)

    def set_codim_nodes(self, ids, mass):
        self.m.set_codim_nodes(ids, mass)

    def set_time_integration(self, *a):
        self.orc.opt_set_time_integration(self.o, *a)

    def enable_self_collision(self, e):
        self.orc.opt_enable_self_collision(self.o, e)

    def add_half_space(self, o, n, e):
        return self.orc.opt_add_half_space(self.o, o, n, e)

    def set_kappa(self, kappa):
        self.o.set_kappa(kappa)

    def set_dhat_target(self, eps):
        self.o.set_dhat_target(eps)

    def set_damping(self, stiff):
        self.o.set_damping(stiff)

    def force_friction_loop(self, on=True):
        self.orc.opt_force_friction_loop(self.o, on)

    def set_friction_scales(self, a, b):
        self.orc.opt_set_friction_scales(self.o, a, b)

    def half_space_move(self, i, delta, slackness=0.5):
        return self.orc.opt_half_space_move(self.o, i, delta, slackness)

    def set_half_space_friction(self, i, mu):
        self.orc.opt_set_half_space_friction(self.o, i, mu)

    def set_friction(self, mu, n, eps):
        self.orc.opt_set_friction(self.o, mu, n, eps)

    def set_friction_target(self, eps):
        self.orc.opt_set_friction_target(self.o, eps)

    def set_parameter_scaling(self, use_abs, dtol_rel, kmm):
        self.orc.opt_set_parameter_scaling(self.o, use_abs, dtol_rel, kmm)

    def add_dirichlet(self, ids, **k):
        self.orc.opt_add_dirichlet(self.o, ids, **k)

    def add_neumann(self, ids, acc, **k):
        self.orc.opt_add_neumann(self.o, ids, acc, **k)

    def end_dirichlet(self, group, t_end):
        self.orc.opt_end_dirichlet(self.o, group, t_end)

    def set_dirichlet_motion(self, group, **k):
        self.orc.opt_set_dirichlet_motion(self.o, group, **k)

    def set_dirichlet_targets(self, group, targets):
        self.orc.opt_set_dirichlet_targets(self.o, group, targets)

    def state(self):
        return self.o.state()

    def solve_timestep(self, cap):
        return self.o.solve_timestep(cap)

    def set_twist(self, l, r):
        self.o.set_twist(l, r)

    def set_velocity(self, v):
        self.orc.opt_set_velocity(self.o, v)

    def set_rel_tol(self, t):
        self.o.set_rel_tol(t)

    def load_status(self, p):
        self.orc.opt_load_status(self.o, p)

    def precompute(self):
        self.o.precompute()


PLANE_OBJ = "v -1 0 -1\nv 1 0 -1\nv 1 0 1\nv -1 0 1\nv 0 0 0\nf 1 5 2\nf 2 5 3\nf 3/1/1 5/2/2 4/3/3\nf -2 -1 1\n"


def test_mesh_obstacles_ride_along_as_surface_only_components(tmp_path):
    """`meshCO` (Config.cpp:448-474, MeshCO.cpp:37-58): centred, rotated, scaled to the given largest extent, moved to the origin;
    appended behind the simulated mesh without tetrahedra, its triangles in the surface, `fall` lifting the simulated mesh only."""
    (tmp_path / "plane.obj").write_text(PLANE_OBJ)
    Vo, Fo = ss.read_obj(tmp_path / "plane.obj")
    assert Vo.shape == (5, 3) and np.array_equal(Fo, [[0, 4, 1], [1, 4, 2], [2, 4, 3], [3, 4, 0]])
    V0, F0 = scene.make_box(1, 1, 1, size=(1.0, 1.0, 1.0), origin=(-0.5, 0.5, -0.5))
    SF0 = scene.surface_tris(F0)
    c = ss.SceneConfig.parse(f"shapes input 1\ncube.msh 0 0 0  0 0 0  1 1 1\nmeshCO {tmp_path}/plane.obj 0.25 -0.5 0  6  50  0.0 rotate 0 90 0\nscript fall\nselfCollisionOff\n")
    sc = ss.assemble(c, lambda p: (V0.copy(), F0.copy(), SF0.copy()))
    n = V0.shape[0]
    assert sc.V.shape[0] == n + 5 and sc.T.shape[0] == F0.shape[0] and np.array_equal(sc.obstacle_nodes, np.arange(n, n + 5))
    P = sc.V[n:]
    assert np.allclose(P.mean(0), [0.25 - 0.0, -0.5, 0.0], atol=1e-12)  # vertex mean at the origin given on the line
    assert abs((P.max(0) - P.min(0)).max() - 6.0) < 1e-12 and np.ptp(P[:, 1]) < 1e-12  # largest extent = scale; still a y = const plane
    assert sc.SF.shape[0] == SF0.shape[0] + 4 and sc.SF[-4:].min() >= n
    lift = 0.5 * np.linalg.norm(V0.max(0) - V0.min(0))
    assert np.allclose(sc.V[:n, 1], V0[:, 1] + lift)  # only the simulated cube is lifted


@pytest.mark.gpu
def test_cube_dropped_on_a_mesh_obstacle_gpu_beside_the_oracle(orc, gpu_lib, tmp_path):
    """A `meshCO` scene end to end (selfCollisionOff: only pairs with the obstacle collide): constraint sets bit-exact while the cube
    lands, converged steps equal, the cube comes to rest on the plane."""
    (tmp_path / "plane.obj").write_text(PLANE_OBJ)
    V, F = scene.make_box(2, 2, 2, size=(0.5, 0.5, 0.5), origin=(-0.25, -0.25, -0.25))
    gl.save_tet_mesh(tmp_path / "cube.msh", scene.jitter(V, F, rel=1e-2), F)
    text = (f"shapes input 1\ncube.msh 0 0.262 0  0 0 0  1 1 1\nmeshCO {tmp_path}/plane.obj 0.1 0 0.05  3  50  0.0 rotate 0 0 0\n"
            "selfCollisionOff\ntime 1 0.01\ntol 1\n1e-6\n")
    cfg = ss.SceneConfig.parse(text, str(tmp_path))
    sc = ss.assemble(cfg, gl.read_tet_mesh)
    ob = ss.apply(sc, OracleBackend(orc))
    gb = ss.apply(sc, gpu_lib.Context(0))
    seen = 0
    for step in range(14):
        no, ng = ob.o.solve_timestep(60), gb.solve_timestep(60)
        assert no < 60 and ng < 60
        so, sg = ob.o.state(), gb.state()
        # dHat = dHatEps^2 * (diagonal of the cube's box)^2: the 3 x 3 obstacle does not count
        assert sg["dHat"] == so["dHat"] and sg["dHat"] == pytest.approx(1e-6 * 0.75, rel=5e-2)
        cs_o, cs_g = orc.opt_contact_state(ob.o), gb.contact_state()
        assert cs_g["nActive"] == len(cs_o["active"]), step
        seen = max(seen, cs_g["nActive"])
        assert np.abs(sg["V"] - so["V"]).max() < 1e-6 * np.abs(so["V"]).max(), step
    n = V.shape[0]
    assert seen > 0 and sg["V"][:n, 1].min() > 0.0  # resting above the plane y = 0
    assert np.array_equal(sg["V"][n:], sc.V[n:])  # the obstacle did not move
    gb.close()


@pytest.mark.gpu
def test_tutorial_scene_runs_on_the_gpu_beside_the_oracle(orc, gpu_lib, tmp_path):
    """input/tutorialExamples/2cubesFall.txt: two cubes, self-contact with friction, rough ground.  The cubes start exactly at
    rest, so single iterates are round-off dependent (makePD2d); the converged steps are compared at a tight tolerance.
    The upper cube is offset and tilted: two exactly aligned cubes are a symmetric configuration whose landing is a bifurcation
    (which corner gives way is decided by round-off and by the adaptive stiffness that follows: a 1e-9 perturbation of the
    CPU restatement's own start moves its own step 7 by 5e-4), so only a generic contact can be compared step by step."""
    V, F = scene.make_box(1, 1, 1, size=(1.0, 1.0, 1.0), origin=(-0.5, -0.5, -0.5))
    gl.save_tet_mesh(tmp_path / "cube.msh", V, F)
    text = TUTORIAL.replace("0 3 0  0 0 0", "0.13 1.66 0.07  4 10 7").replace("0 1 0", "0 0.52 0") + "tol 1\n1e-6\n"  # closer to the ground: contact within a few steps
    cfg = ss.SceneConfig.parse(text, str(tmp_path))
    sc = ss.assemble(cfg, gl.read_tet_mesh)
    assert sc.V.shape == (16, 3) and sc.T.shape == (12, 4)
    ob = ss.apply(sc, OracleBackend(orc))
    c = ss.apply(sc, gpu_lib.Context(0))
    touched = pairs = 0
    for step in range(14):
        no = 0
        ob.o.begin_timestep()
        c.begin_timestep()
        for sub in range(4):
            for it in range(200):
                co, cg = ob.o.newton_iter(), c.newton_iter()
                if co and cg:
                    break
            else:
                pytest.fail("no convergence")
            mo, mg = orc.opt_next_subproblem(ob.o), c.next_subproblem()
            assert mo == mg
            if not mo:
                break
        ob.o.end_timestep()
        c.end_timestep()
        so, sg = ob.o.state(), c.state()
        assert np.abs(sg["V"] - so["V"]).max() < 1e-6, step
        touched += c.contact_state()["nHalfSpace"] > 0
        pairs += c.contact_state()["nActive"] > 0
    assert pairs >= 3  # the cubes are in contact with each other
    assert touched >= 3 and c.state()["V"][:, 1].min() > 0.0  # the lower cube has reached the ground and stays above it
    c.close()


def _intersecting_start(tmp_path):
    V, F = scene.make_box(1, 1, 1, size=(1.0, 1.0, 1.0), origin=(-0.5, -0.5, -0.5))
    gl.save_tet_mesh(tmp_path / "cube.msh", V, F)
    cfg = ss.SceneConfig.parse(TUTORIAL.replace("0 3 0  0 0 0", "0.13 1.45 0.07  4 10 7"), str(tmp_path))  # a corner of the upper cube inside the lower
    return ss.assemble(cfg, gl.read_tet_mesh)


def test_intersecting_start_is_refused(orc, tmp_path):
    """Optimizer.cpp:258-263: "intersection detected in initial configuration!" ends the reference's process."""
    with pytest.raises(RuntimeError, match="intersection detected in initial configuration"):
        ss.apply(_intersecting_start(tmp_path), OracleBackend(orc))


@pytest.mark.gpu
def test_intersecting_start_is_refused_on_the_gpu(gpu_lib, tmp_path):
    c = gpu_lib.Context(0)
    with pytest.raises(gl.IpcGpuError, match="intersection detected in initial configuration"):
        ss.apply(_intersecting_start(tmp_path), c)
    c.close()


def test_shape_lines_continue_over_backslashes():
    """tutorialExamples/BC/2cubesFall_DBC_timeRange.txt: a shape whose DBC entries sit on continuation lines (Config.cpp:290-305)."""
    text = ("shapes input 2\na.msh 0 3 0  0 0 0  1 1 1\n"
            "a.msh 0 1 0  0 0 0  1 1 1 \\\n    DBC -0.1 -0.1 -0.1  0.1 1.1 0.1  -0.2 0.0 -0.2  0 0 0  0.0 2.5 \\\n"
            "    DBC 0.9 -0.1 0.9  1.1 1.1 1.1  0.2 0 0.2  0 0 0  2.5   # from 2.5 s on \\\n    NBC 0 0 0 1 1 1  0 -1 0\nselfFric 0.1\n")
    c = ss.SceneConfig.parse(text)
    assert len(c.shapes) == 2 and not c.shapes[0].dbc and c.self_fric == 0.1
    d = c.shapes[1].dbc
    assert len(d) == 2 and d[0][4:] == (0.0, 2.5) and d[1][4] == 2.5 and d[1][5] == float("inf") and len(c.shapes[1].nbc) == 1


def test_dragright_grabs_the_right_end_and_lets_go_behind_the_obstacles(orc, tmp_path):
    """`script dragright` (AnimScripter.cpp:809-826, 1619-1632): body lifted like `fall`, the rightmost 4 % of the nodes pulled at
    0.5 in +x as a NONZERO handle; once every node is right of the obstacles the handle is released and the body falls freely."""
    (tmp_path / "plane.obj").write_text(PLANE_OBJ)
    V0, F0 = scene.make_box(4, 1, 1, size=(1.0, 0.25, 0.25), origin=(0.0, 0.0, 0.0))
    SF0 = scene.surface_tris(F0)
    text = (f"script dragright\nshapes input 1\nbar.msh 0 0 0  0 0 0  1 1 1\nmeshCO {tmp_path}/plane.obj -0.75 -2 0  0.5  50  0.0\n"
            "selfCollisionOff\ntime 1 0.02\ntol 1\n1e-6\n")
    cfg = ss.SceneConfig.parse(text, str(tmp_path))
    sc = ss.assemble(cfg, lambda p: (V0.copy(), F0.copy(), SF0.copy()))
    n = V0.shape[0]
    lift = 0.5 * np.linalg.norm(V0.max(0) - V0.min(0))
    assert np.allclose(sc.V[:n, 1], V0[:, 1] + lift)
    (ids, lin, ang, t0, t1), = sc.dirichlet
    assert np.array_equal(ids, np.nonzero(V0[:, 0] > 0.96)[0]) and lin == (0.5, 0.0, 0.0) and t1 == float("inf")
    assert sc.release["x_limit"] == pytest.approx(-0.5) and sc.release["nSim"] == n  # obstacle: x in [-1, -0.5]
    be = ss.apply(sc, OracleBackend(orc))
    # the body starts at x in [0, 1], right of the obstacle: the very first stepAnimScript lets go
    x0 = be.state()["V"][ids, 0].copy()
    assert sc.before_step(be, 0.0) and not sc.before_step(be, 0.02)
    assert be.solve_timestep(50) < 50
    s1 = be.state()
    # nobody pulls any more (a pulled handle would be 0.5 dt = 1e-2 further right); the body falls.  As in the reference the
    # released nodes still carry the xTilde of a Dirichlet node for this one step (computeXTilta ran before stepAnimScript let go,
    # Optimizer.cpp:524-530, 1247-1249), so they lag behind the free fall g dt^2 of the others
    assert np.abs(s1["V"][ids, 0] - x0).max() < 2e-3
    fall = sc.V[:n, 1] - s1["V"][:n, 1]
    assert np.all(fall > 0) and fall.max() == pytest.approx(9.80665 * 0.02 ** 2, rel=2e-2) and fall[ids].max() < 0.5 * fall.max()
    # with the obstacle to the right the handle keeps pulling: 0.5 * dt per step, exactly
    sc2 = ss.assemble(ss.SceneConfig.parse(text.replace("-0.75 -2 0", "3 -2 0"), str(tmp_path)), lambda p: (V0.copy(), F0.copy(), SF0.copy()))
    be2 = ss.apply(sc2, OracleBackend(orc))
    for k in range(2):
        assert not sc2.before_step(be2, 0.02 * k)
        assert be2.solve_timestep(50) < 50
    assert np.allclose(be2.state()["V"][ids, 0], x0 + 2 * 0.5 * 0.02, atol=1e-12)


@pytest.mark.gpu
def test_dragright_on_the_gpu_beside_the_oracle(orc, gpu_lib, tmp_path):
    """The pulled handle and its state-dependent release through the C ABI (ipcgpu_opt_end_dirichlet): same decisions, same steps."""
    (tmp_path / "plane.obj").write_text(PLANE_OBJ)
    V0, F0 = scene.make_box(4, 1, 1, size=(1.0, 0.25, 0.25), origin=(0.0, 0.0, 0.0))
    V0 = scene.jitter(V0, F0, rel=1e-2)
    SF0 = scene.surface_tris(F0)
    # the obstacle's right end is at x = 0.015
    text = (f"script dragright\nshapes input 1\nbar.msh 0 0 0  0 0 0  1 1 1\nmeshCO {tmp_path}/plane.obj -0.235 -2 0  0.5  50  0.0\n"
            "selfCollisionOff\ntime 1 0.02\ntol 1\n1e-6\n")
    cfg = ss.SceneConfig.parse(text, str(tmp_path))
    read = lambda p: (V0.copy(), F0.copy(), SF0.copy())
    sco, scg = ss.assemble(cfg, read), ss.assemble(cfg, read)
    assert sco.release["x_limit"] == pytest.approx(0.015)
    ob, gb = ss.apply(sco, OracleBackend(orc)), ss.apply(scg, gpu_lib.Context(0))
    released = []
    for k in range(8):
        ro, rg = sco.before_step(ob, 0.02 * k), scg.before_step(gb, 0.02 * k)
        assert ro == rg
        released.append(ro)
        no, ng = ob.solve_timestep(60), gb.solve_timestep(60)
        assert no < 60 and ng == no
        so, sg = ob.state(), gb.state()
        assert np.abs(sg["V"] - so["V"]).max() < 1e-8 * np.abs(so["V"]).max(), k
    assert released == [False] * 5 + [True, False, False]  # the left end follows the pulled handle elastically: past x = 0.015 after five steps
    gb.close()


DCOFIX = ("script DCOFix\nshapes input 2\n{plane} 0 -0.01 0  0 0 0  3 1 3\ncube.msh 0 0.012 0  0 0 0  1 1 1\n"
          "selfCollisionOn\ntime 1 0.01\ntol 1\n1e-6\n")


def test_triangle_meshes_under_shapes_are_surface_only_components_of_the_mesh(tmp_path):
    """`.obj` under `shapes` (main.cpp:948-956, Mesh.cpp:310-345): nodes of no tetrahedron that still belong to Mesh<3> -- lumped
    masses from the triangle areas; `script DCOFix` holds them; `shapeMatrix` replicates a shape."""
    (tmp_path / "plane.obj").write_text(PLANE_OBJ)
    V0, F0 = scene.make_box(1, 1, 1, size=(0.5, 0.5, 0.5), origin=(-0.25, 0.0, -0.25))
    SF0 = scene.surface_tris(F0)
    cfg = ss.SceneConfig.parse(DCOFIX.format(plane=tmp_path / "plane.obj"), str(tmp_path))
    sc = ss.assemble(cfg, lambda p: (V0.copy(), F0.copy(), SF0.copy()))
    assert np.array_equal(sc.codim_nodes, np.arange(5)) and np.array_equal(sc.codim_fixed, sc.codim_nodes) and sc.obstacle_nodes is None
    assert sc.T.min() >= 5 and sc.SF.shape[0] == 4 + SF0.shape[0] and not sc.dirichlet
    # the plane is 2 x 2 scaled by 3: area 36, four triangles of area 9; corner nodes touch two of them, the centre all four
    assert np.allclose(sc.codim_mass, 1000.0 * np.array([6.0, 6.0, 6.0, 6.0, 12.0]))
    m = ss.SceneConfig.parse("shapeMatrix input 2 1 3  1 2 3\na.obj 10 0 5  0 0 0  1 1 1\n")
    assert len(m.shapes) == 6 and [tuple(x.translate) for x in m.shapes][:4] == [(1, 2, 3), (1, 2, 8), (1, 2, 13), (11, 2, 3)]
    # `.seg` / `.pt` shapes (main.cpp:957-1005): segments from the file or from the triangles of the .obj beside it, points from its vertices;
    # a segment end carries density * l^3 pi / 12, a point the mean nodal mass of the tetrahedral components (Mesh.cpp:279-295, 405-411)
    (tmp_path / "rope.seg").write_text("v 0 0 0\nv 2 0 0\nv 2 1 0\ns 1 2\ns 2 3\n")
    Vs, Es = ss.read_seg(str(tmp_path / "rope.seg"))
    assert Vs.shape == (3, 3) and Es.tolist() == [[0, 1], [1, 2]]
    Vt, Et = ss.read_seg(str(tmp_path / "plane.seg"))  # no such file: the edges of plane.obj, each once, in std::set order
    assert Vt.shape == (5, 3) and len(Et) == 8 and Et.tolist() == sorted(Et.tolist()) and len({frozenset(e) for e in Et.tolist()}) == 8
    assert ss.read_pt(str(tmp_path / "plane.pt")).shape == (5, 3)
    txt = ("script null\nshapes input 3\nbox.msh 0 2 0  0 0 0  1 1 1\n" + str(tmp_path / "rope.seg") + " 0 0 0  0 0 0  1 1 1 linearVelocity 0 0 0\n"
           + str(tmp_path / "plane.pt") + " 0 -1 0  0 0 0  1 1 1 angularVelocity 0 10 0\n")
    sc2 = ss.assemble(ss.SceneConfig.parse(txt, str(tmp_path)), lambda p: (V0.copy(), F0.copy(), SF0.copy()))
    nb = V0.shape[0]
    assert sc2.codim_edges.tolist() == [[nb, nb + 1], [nb + 1, nb + 2]] and np.array_equal(sc2.codim_nodes, np.arange(nb, nb + 8))
    assert np.allclose(sc2.codim_mass[:3], 1000.0 * np.pi / 12.0 * np.array([8.0, 8.0 + 1.0, 1.0]))
    assert np.allclose(sc2.codim_mass[3:], 1000.0 * 0.125 / nb)  # the box of volume 1/8 over its nb nodes
    assert len(sc2.dirichlet) == 2 and sc2.SF.shape[0] == SF0.shape[0]
    with pytest.raises(ss.UnsupportedKeyword):  # nobody holds the surface: it would fall as a cloud of free particles
        ss.assemble(ss.SceneConfig.parse(DCOFIX.format(plane=tmp_path / "plane.obj").replace("script DCOFix\n", ""), str(tmp_path)),
                    lambda p: (V0.copy(), F0.copy(), SF0.copy()))


def test_cube_lands_on_a_fixed_surface_of_the_mesh_oracle(orc, tmp_path):
    """DCOFix end to end on the oracle: the held surface takes part in self-collision; dHat and kappa are sized by the tetrahedral
    components alone (Optimizer.cpp:101, 2220: matSpaceBBoxSize2(dim), avgNodeMass(dim) -- checked against a run of the
    reference-compiled code, tools/ref_compare.py on 2cubesFall_rotateCO_closedSurface.txt)."""
    (tmp_path / "plane.obj").write_text(PLANE_OBJ)
    V0, F0 = scene.make_box(2, 2, 2, size=(0.5, 0.5, 0.5), origin=(-0.25, 0.0, -0.25))
    SF0 = scene.surface_tris(F0)
    cfg = ss.SceneConfig.parse(DCOFIX.format(plane=tmp_path / "plane.obj"), str(tmp_path))
    sc = ss.assemble(cfg, lambda p: (V0.copy(), F0.copy(), SF0.copy()))
    be = ss.apply(sc, OracleBackend(orc))
    diag2 = float(((sc.V[5:].max(0) - sc.V[5:].min(0)) ** 2).sum())  # the cube alone, not the 6 x 6 plane
    assert be.state()["dHat"] == pytest.approx(1e-6 * diag2, rel=1e-12)
    seen = 0
    for k in range(12):
        assert be.solve_timestep(80) < 80
        seen = max(seen, len(orc.opt_contact_state(be.o)["active"]))
    s = be.state()
    assert seen > 0 and s["V"][5:, 1].min() > -0.01  # resting on the plane y = -0.01
    assert np.array_equal(s["V"][:5], sc.V[:5])  # which did not move


@pytest.mark.gpu
def test_dcofix_scene_on_the_gpu_beside_the_oracle(orc, gpu_lib, tmp_path):
    (tmp_path / "plane.obj").write_text(PLANE_OBJ)
    V0, F0 = scene.make_box(2, 2, 2, size=(0.5, 0.5, 0.5), origin=(-0.25, 0.0, -0.25))
    V0 = scene.jitter(V0, F0, rel=1e-2)
    SF0 = scene.surface_tris(F0)
    cfg = ss.SceneConfig.parse(DCOFIX.format(plane=tmp_path / "plane.obj"), str(tmp_path))
    read = lambda p: (V0.copy(), F0.copy(), SF0.copy())
    ob, gb = ss.apply(ss.assemble(cfg, read), OracleBackend(orc)), ss.apply(ss.assemble(cfg, read), gpu_lib.Context(0))
    seen = 0
    for k in range(12):
        no, ng = ob.solve_timestep(80), gb.solve_timestep(80)
        assert no < 80 and ng < 80
        so, sg = ob.state(), gb.state()
        assert sg["dHat"] == so["dHat"] and sg["kappa"] == pytest.approx(so["kappa"], rel=1e-7)  # kappa_bal = -g_c.g_E / |g_c|^2 cancels: a ratio that amplifies round-off
        cs_o, cs_g = orc.opt_contact_state(ob.o), gb.contact_state()
        assert cs_g["nActive"] == len(cs_o["active"]), k
        seen = max(seen, cs_g["nActive"])
        assert np.abs(sg["V"] - so["V"]).max() < 1e-6 * np.abs(so["V"]).max(), k
    assert seen > 0
    gb.close()


ACO6 = """energy NH
time 1 0.025
density 1000
stiffness 1e5 0.4
script ACOSquash6
shapes input 1
box.msh 0 0 0  0 0 0  1 1 1
selfCollisionOff
tol 1
1e-4
halfSpace -0.2 0.5 0.5  1 0 0  1000 0
halfSpace 1.2 0.5 0.5  -1 0 0  1000 0
halfSpace 0.5 -0.2 0.5  0 1 0  1000 0
halfSpace 0.5 1.2 0.5  0 -1 0  1000 0
halfSpace 0.5 0.5 -0.2  0 0 1  1000 0
halfSpace 0.5 0.5 1.2  0 0 -1  1000 0
"""


def test_aco_squash_rules():
    """ACOSquash6 (AnimScripter.cpp:966-985, 1848-1871): six planes closing at 1, each move cut short by HalfSpace::move's slackness; the tooling
    keeps the origins it asked the backend to move to."""
    cfg = ss.SceneConfig.parse(ACO6, "/x")
    assert cfg.script == "ACOSquash6" and len(cfg.half_spaces) == 6
    V0, F0 = scene.make_box(1, 1, 1)
    read = lambda p: (V0.copy(), F0.copy(), scene.surface_tris(F0))
    sc = ss.assemble(cfg, read)
    assert sc.release["kind"] == "aco" and len(sc.release["vel"]) == 6 and sc.dirichlet == []
    assert np.array_equal(sc.release["vel"][3], [0.0, -1.0, 0.0]) and sc.release["pairs"] == [(0, 0.2), (2, 0.2), (4, 0.2)]
    with pytest.raises(ss.UnsupportedKeyword):
        ss.assemble(ss.SceneConfig.parse(ACO6.replace("halfSpace 0.5 0.5 1.2  0 0 -1  1000 0\n", ""), "/x"), read)


@pytest.mark.gpu
def test_aco_squash6_on_the_gpu_beside_the_oracle(orc, gpu_lib):
    """The planes of `script ACOSquash6` moved by ipcgpu_halfspace_move before every step: the unit cube is caught by the rising bottom plane and
    boxed in; plane moves (fractions left), constraint sets and iterates next to the CPU restatement."""
    V0, F0 = scene.make_box(2, 2, 2, size=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0))
    V0 = scene.jitter(V0, F0, rel=1e-2)
    SF0 = scene.surface_tris(F0)
    cfg = ss.SceneConfig.parse(ACO6, "/x")
    read = lambda p: (V0.copy(), F0.copy(), SF0.copy())
    so_, sg_ = ss.assemble(cfg, read), ss.assemble(cfg, read)
    ob, gb = ss.apply(so_, OracleBackend(orc)), ss.apply(sg_, gpu_lib.Context(0))
    limited = False
    for k in range(16):
        so_.before_step(ob, k * cfg.dt)
        sg_.before_step(gb, k * cfg.dt)
        for a, b in zip(so_.release["origins"], sg_.release["origins"]):
            assert np.abs(a - b).max() <= 1e-9
        limited |= abs(so_.release["origins"][2][1] - (-0.2 + (k + 1) * cfg.dt)) > 1e-6  # the bottom plane no longer takes its whole step
        no, ng = ob.solve_timestep(200), gb.solve_timestep(200)
        assert no < 200 and ng == no, (k, no, ng)
        assert np.abs(gb.state()["V"] - ob.state()["V"]).max() < 1e-7 * np.abs(ob.state()["V"]).max(), k
    assert limited
    gb.close()


class _MockBackend:
    """records what AssembledScene.before_step asks for; positions are set by the test"""

    def __init__(self, V):
        self.V, self.calls = np.array(V, dtype=np.float64), []

    def state(self):
        return {"V": self.V}

    def set_dirichlet_motion(self, g, **k):
        self.calls.append(("motion", g, tuple(k.get("lin_vel", (0, 0, 0))), tuple(k.get("ang_vel_deg", (0, 0, 0)))))

    def end_dirichlet(self, g, t):
        self.calls.append(("end", g, t))


def _box_scene(script, n=(2, 2, 2), second=None):
    V0, F0 = scene.make_box(*n)
    shapes = "m.msh 0 0 0  0 0 0  1 1 1\n" + (second or "")
    cfg = ss.SceneConfig.parse(f"script {script}\nshapes input {1 + (second is not None)}\n{shapes}", "/x")
    return ss.assemble(cfg, lambda p: (V0.copy(), F0.copy(), scene.surface_tris(F0)))


def test_before_step_rules_on_a_mock_backend():
    """The state-dependent halves of the rule scripts (AnimScripter::stepAnimScript), driven by hand: `push` stops for good, `tear` changes sign in
    EVERY step that finds the turning node beyond its mark, `toggleTop` lets its set go once, `DCOCut` pauses while the knife is below 0.001."""
    sc = _box_scene("push")
    be = _MockBackend(sc.V)
    r = sc.release
    assert r["kind"] == "turn" and r["stop"] and not sc.before_step(be, 0.0) and be.calls == []
    be.V[r["turn"], 1] = r["lo"] - 1e-3
    assert sc.before_step(be, 0.1) and be.calls[-1] == ("motion", 1, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)) and r["done"]
    assert not sc.before_step(be, 0.2)  # nothing more, whatever the node does
    sc = _box_scene("tear")
    be, r = _MockBackend(sc.V), sc.release
    be.V[r["turn"], 0] = r["lo"] - 1e-3
    assert sc.before_step(be, 0.0) and be.calls[-1][2] == (5.0, 0.0, 0.0)
    assert sc.before_step(be, 0.1) and be.calls[-1][2] == (-5.0, 0.0, 0.0)  # still beyond the mark: the sign changes again
    be.V[r["turn"], 0] = r["lo"] + 1.0
    assert not sc.before_step(be, 0.2)
    sc = _box_scene("toggleTop")
    be, r = _MockBackend(sc.V), sc.release
    assert r["kind"] == "let_go" and not sc.before_step(be, 0.0)
    be.V[r["turn"], 0] = r["limit"] - 1e-6
    assert sc.before_step(be, 0.3) and be.calls == [("end", 0, 0.3)] and not sc.before_step(be, 0.4)
    sc = _box_scene("DCOCut", second="k.msh 0 2 0  0 0 0  1 1 1\n")
    be, r = _MockBackend(sc.V), sc.release
    assert r["kind"] == "while_above" and len(sc.dirichlet) == 1 and sc.motions[0][0] == (0.0, -1.0, -1.0) and not sc.before_step(be, 0.0)
    be.V[r["ids"], 1] -= be.V[r["ids"], 1].min() - 5e-4  # the knife's lowest node below the mark
    assert sc.before_step(be, 0.1) and be.calls[-1][:3] == ("motion", 0, (0.0, 0.0, 0.0))
    be.V[r["ids"], 1] += 0.1
    assert sc.before_step(be, 0.2) and be.calls[-1][:3] == ("motion", 0, (0.0, -1.0, -1.0))
