"""The reference's OWN main() on the HIP path: tests/adapters/_build/ipc_main_hip is src/main.cpp compiled where it lies (its one
`new Optimizer<DIM>` at :1397 redirected to include/adapters/HipOptimizer.hpp by the pre-included tests/adapters/main_hook.hpp),
linked with the reference's other compiled sources and libipcgpu.so (tests/test_adapters.py::build_main_hip; built in the
container that holds /root/reference, the executable travels to the GPU box).  Scene scripts the reference itself ran
(tests/golden/ref_scene_*.npz: positions after every step, Newton iterations per step) are exported to a scratch directory -- script
+ mesh files, written from the fixture -- and run through that executable in offline mode (`ipc_main_hip 100 scene.txt -o out/`),
exactly as a user runs IPC_bin; info<N>.txt / status<N> are read back like tools/ref_compare.py reads the reference's."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from test_adapters import LIB_REF, MAIN_HIP, REF_SRC  # noqa: E402
from test_oracle_vs_reference import load_scene  # noqa: E402


def export_scene(S, meshes, out_dir, steps):
    """Script + every mesh file it names, from the fixture; returns the script path.  `time` is overridden to `steps` steps."""
    from ipc_amd import lib as gl
    from ipc_amd import scene_script as ss
    text = str(S["script"])
    cfg = ss.SceneConfig.parse(text, out_dir)
    for key, (V, T, SF, E) in meshes.items():
        path = os.path.join(out_dir, key)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        if key.lower().endswith(".seg"):
            with open(path, "w") as f:  # IglUtils::readSEG, IglUtils.cpp:146-175
                for v in V:
                    f.write("v %.17g %.17g %.17g\n" % tuple(v))
                for e in E:
                    f.write("s %d %d\n" % tuple(int(i) + 1 for i in e))
        elif key.lower().endswith(".pt"):
            with open(path, "w") as f:
                for v in V:
                    f.write("v %.17g %.17g %.17g\n" % tuple(v))
        elif key.lower().endswith(".obj"):
            with open(path, "w") as f:
                for v in V:
                    f.write("v %.17g %.17g %.17g\n" % tuple(v))
                for t in SF:
                    f.write("f %d %d %d\n" % tuple(int(i) + 1 for i in t))
        else:
            gl.save_tet_mesh(path, V, T)
    lines = [ln for ln in text.splitlines() if not ln.strip().startswith("time ")]
    script = os.path.join(out_dir, "scene.txt")
    with open(script, "w") as f:
        f.write("\n".join(lines) + f"\ntime {steps * cfg.dt:.17g} {cfg.dt:.17g}\n")
    return script


def read_run(out_dir, steps):
    import ref_compare as rc
    its = rc.read_iter_counts(out_dir, steps)
    pos = np.array([rc.read_status_positions(os.path.join(out_dir, f"status{s + 1}")) for s in range(steps)])
    return pos, its


def run_main_hip(S, meshes, tmp, steps, mode="resident", extra_env=None):
    script = export_scene(S, meshes, str(tmp), steps)
    out = os.path.join(str(tmp), "out_" + mode)
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, IPCGPU_OPTIMIZER_MODE=mode, **(extra_env or {}))
    r = subprocess.run([MAIN_HIP, "100", script, "-o", out + "/", "--logLevel", "off"], cwd=str(tmp), env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=1500)
    log = r.stdout.decode(errors="replace")
    assert r.returncode == 0, log[-3000:]
    return read_run(out, steps) + (log,)


needs_exe = pytest.mark.skipif(not os.path.exists(MAIN_HIP), reason="tests/adapters/_build/ipc_main_hip is built in the container that holds /root/reference")


@pytest.mark.skipif(not (os.path.isdir(REF_SRC) and os.path.exists(LIB_REF)), reason="needs the reference-compiled library (build container)")
def test_exported_scene_reproduces_the_fixture_through_the_reference(tmp_path):
    """The export itself (script + mesh files written from the fixture) is faithful: the reference-compiled library, run on the exported
    directory, returns what it returned on its own input files."""
    import ref_compare as rc
    S, meshes = load_scene("two_cubes_fall")
    steps = 20
    script = export_scene(S, meshes, str(tmp_path), steps)
    rcode, log = rc.run_reference(script, os.path.join(str(tmp_path), "ref"), cwd=str(tmp_path))
    assert rcode == 0, log[-2000:]
    pos, its = read_run(os.path.join(str(tmp_path), "ref"), steps)
    assert np.array_equal(its, S["iters"][:steps])
    assert np.abs(pos - S["positions"][:steps]).max() <= 1e-9


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="build container only")
def test_main_hip_builds():
    import test_adapters
    exe = test_adapters.build_main_hip()
    assert os.path.exists(exe)


@pytest.mark.gpu
@needs_exe
def test_reference_main_bar_twist_resident(tmp_path):
    """BASELINE configs[0] (barTwist_noCollisions.txt) through the reference's main() with HipOptimizer in resident mode: the reference's
    Newton iteration count in every step, positions within the Newton tolerance of the script (the criteria of
    test_gpu_vs_reference.py::test_scene_bar_twist_against_the_reference, which drives the same library from Python)."""
    S, meshes = load_scene("bar_twist")
    pos, its, _ = run_main_hip(S, meshes, tmp_path, 4)
    assert np.array_equal(its, S["iters"][:4]), (its.tolist(), S["iters"].tolist())
    for s in range(4):
        assert np.abs(pos[s] - S["positions"][s]).max() <= 1e-5 * np.abs(S["positions"][s]).max()


@pytest.mark.gpu
@needs_exe
def test_reference_main_two_cubes_fall_resident(tmp_path):
    """tutorialExamples/2cubesFall.txt (ground with friction, self-contact) through the reference's main() on the HIP path: free fall
    identical to round-off, the reference's iteration counts through both impacts except around the first touch-down (F = I up to
    round-off there: IglUtils::makePD2d's discontinuity) -- the criteria of test_scene_two_cubes_fall_against_the_reference."""
    S, meshes = load_scene("two_cubes_fall")
    steps = int(S["steps"])
    pos, its, _ = run_main_hip(S, meshes, tmp_path, steps)
    free = 17
    for s in range(free):
        assert np.abs(pos[s] - S["positions"][s]).max() <= 1e-12
    assert np.array_equal(its[:free], S["iters"][:free])
    differ = np.nonzero(its != S["iters"])[0]
    assert len(differ) <= 8, (its.tolist(), S["iters"].tolist())
    assert abs(int(its.sum()) - int(S["iters"].sum())) <= 0.12 * int(S["iters"].sum())
    assert np.abs(pos[-1] - S["positions"][-1]).max() <= 1e-2 * np.abs(S["positions"][-1]).max()


@pytest.mark.gpu
@needs_exe
def test_reference_main_matches_the_python_driven_stepper(tmp_path, gpu_lib):
    """The adapter hands the library what ipc_amd/scene_script.py hands it: a scene with a kinematic mesh obstacle, scripted component
    velocities and contact (2cubesFall_rotateCO.txt) gives the same trajectory either way."""
    from test_oracle_vs_reference import run_scene
    S, meshes = load_scene("rotate_co")
    steps = 12
    pos, its, _ = run_main_hip(S, meshes, tmp_path, steps)
    c = gpu_lib.Context(0)
    pos_py, its_py = run_scene(S, meshes, c, steps)
    c.close()
    assert np.array_equal(its, its_py), (its.tolist(), its_py.tolist())
    n = pos.shape[1]
    assert np.abs(pos - pos_py[:, :n]).max() <= 1e-8 * np.abs(pos_py).max()
    assert np.array_equal(its, S["iters"][:steps]), (its.tolist(), S["iters"][:steps].tolist())


@pytest.mark.gpu
@needs_exe
def test_reference_main_bar_twist_percall(tmp_path):
    """percall mode: the reference's own solve() / fullyImplicit / lineSearch with HipElasticEnergy as its elasticity term and
    HipLinSysSolver from LinSysSolver::create (A/B inside one binary).  Everything but the element kernels and the Cholesky is the
    reference's arithmetic, so the run tracks the reference's: same counts, positions as above."""
    S, meshes = load_scene("bar_twist")
    pos, its, _ = run_main_hip(S, meshes, tmp_path, 3, mode="percall")
    assert np.array_equal(its, S["iters"][:3]), (its.tolist(), S["iters"].tolist())
    for s in range(3):
        assert np.abs(pos[s] - S["positions"][s]).max() <= 1e-5 * np.abs(S["positions"][s]).max()


@pytest.mark.gpu
@needs_exe
def test_reference_main_two_cubes_fall_percall_contact_on_the_device(tmp_path):
    """Round 5, percall mode with self-contact ON THE DEVICE: the reference's own solve() / fullyImplicit_IP() / solveSub_IP() / lineSearch() drive the
    loop, and the 44 `SelfCollisionHandler<dim>::` call sites of its UNCHANGED Optimizer.cpp (compiled with tests/adapters/optimizer_hook.hpp pre-included)
    reach the statics of include/adapters/HipSelfCollisionHandler.hpp: constraint sets, per-constraint distances and Jacobian products, the barrier Hessian
    added in HBM, both CCD step bounds and the intersection test run through the C ABI; the ground plane, friction and the mollified pairs stay host code.
    The run says which mode it is in and how many calls went to the device; it is held to the same criterion as the resident stepper (the envelope of the
    reference's own one-ulp ensemble from the first touch-down on)."""
    from test_oracle_vs_reference import check_envelope
    S, meshes = load_scene("two_cubes_fall")
    steps = 25
    pos, its, log = run_main_hip(S, meshes, tmp_path, steps, mode="percall")
    assert "self-contact on the device" in log, log[-1500:]
    import re
    m = re.search(r"forwarded to the device: (\d+) constraint sets, (\d+) evaluations, (\d+) Jacobian products, (\d+) barrier Hessians, (\d+) \+ (\d+) step bounds, "
                  r"(\d+) intersection tests \((\d+) Hessians fell back", log)
    assert m, log[-1500:]
    n = [int(x) for x in m.groups()]
    assert min(n[0], n[1], n[2], n[3], n[4], n[6]) > 0 and n[7] == 0, n  # every kind of call reached the device (the full sweep n[5] only runs when a step leaves the CFL ball)
    check_envelope("two_cubes_fall", S, pos, its)


@pytest.mark.gpu
@needs_exe
def test_reference_main_two_cubes_fall_percall(tmp_path):
    """percall mode with contact ON THE HOST (IPCGPU_PERCALL_CONTACT=host, the A/B of the test above and all there was before round 5): the barrier terms,
    constraint sets and CCD are the reference's host code adding into the HipLinSysSolver's pending host updates; elasticity and the factorisation run on
    the device."""
    S, meshes = load_scene("two_cubes_fall")
    steps = 25
    pos, its, log = run_main_hip(S, meshes, tmp_path, steps, mode="percall", extra_env={"IPCGPU_PERCALL_CONTACT": "host"})
    assert "contact on the host" in log, log[-1500:]
    free = 17
    for s in range(free):
        assert np.abs(pos[s] - S["positions"][s]).max() <= 1e-12
    assert np.array_equal(its[:free], S["iters"][:free])
    assert abs(int(its.sum()) - int(S["iters"][:steps].sum())) <= 0.15 * int(S["iters"][:steps].sum()), (its.tolist(), S["iters"].tolist())


@pytest.mark.gpu
@needs_exe
def test_reference_main_squash6_resident(tmp_path):
    """`script DCOSquash6` (six closing plates, FCR) through the reference's main() with HipOptimizer resident: the adapter evaluates the
    script's rule (AnimScripter.cpp:2053-2074) before every step and hands the plate motion to ipcgpu_opt_set_dirichlet_motion."""
    from test_oracle_vs_reference import check_plates
    S, meshes = load_scene("squash6_contact")
    pos, its, _ = run_main_hip(S, meshes, tmp_path, int(S["steps"]))
    check_plates(S, pos, its, 1e-6)


@pytest.mark.gpu
@needs_exe
@pytest.mark.parametrize("name,tol", [("rotate_co_edges", 1e-6), ("rotate_co_points", 1e-5)])
def test_reference_main_codimensional_shapes_resident(name, tol, tmp_path):
    """`.seg` / `.pt` shapes read by the reference's own main() (readSEG / the vertices of an .obj, main.cpp:957-1005) and handed to the
    library by HipOptimizer (Mesh<3>::CE through ipcgpu_set_surface_codim, masses through ipcgpu_set_mesh_features): all 44 Newton counts."""
    from test_oracle_vs_reference import check_codim
    S, meshes = load_scene(name)
    pos, its, _ = run_main_hip(S, meshes, tmp_path, int(S["steps"]))
    check_codim(S, pos, its, tol)


@pytest.mark.gpu
@needs_exe
def test_reference_main_absolute_parameters_resident(tmp_path):
    """`useAbsParameters`, `kappaMinMultiplier` and a six-entry `tuning` read by the reference's Config and handed over by HipOptimizer
    (ipcgpu_opt_set_parameter_scaling, ipcgpu_opt_set_dhat_target): the Newton counts of the reference's own run, through the impact."""
    S, meshes = load_scene("point_triangle_abs_parameters")
    steps = 30
    pos, its, log = run_main_hip(S, meshes, tmp_path, steps)
    assert "percall mode" not in log
    assert np.array_equal(its, S["iters"][:steps]), (its.tolist(), S["iters"][:steps].tolist())
    assert np.abs(pos[-1] - S["positions"][steps - 1]).max() <= 1e-5 * np.abs(S["positions"][steps - 1]).max()  # (Newton tolerance 2e-2 absolute; observed 1.4e-6)


@pytest.mark.gpu
@needs_exe
@pytest.mark.parametrize("name", ["script_hang2", "script_corner", "script_left_hit_right", "script_stamp_inv", "script_squash", "script_dragdown"])
def test_reference_main_static_scripts_resident(name, tmp_path):
    """Scripts whose effect is decided in the base-class constructor (held node sets ZERO / NONZERO, start velocities, changed start positions) and
    sets pulled at a constant velocity (squash; dragdown through the barrier of a ground plane): HipOptimizer hands the sets the reference's own
    AnimScripter picked to the library and stays in resident mode."""
    from test_oracle_vs_reference import check_boxrule
    S, meshes = load_scene(name)
    pos, its, log = run_main_hip(S, meshes, tmp_path, int(S["steps"]))
    assert "percall mode" not in log
    if name == "script_stamp_inv":  # 548 Newton iterations out of an inside-out start (see test_gpu_vs_reference.py)
        assert abs(int(its[0]) - int(S["iters"][0])) <= 30 and np.abs(pos[-1] - S["positions"][-1]).max() <= 1e-5 * np.abs(S["positions"]).max(), its.tolist()
        return
    # (the first step starts from the exact rest shape: its count may straddle the tolerance -- corner takes 3 iterations here, 4 in the reference)
    assert np.array_equal(its[1:], S["iters"][1:]) and abs(int(its[0]) - int(S["iters"][0])) <= 1, (its.tolist(), S["iters"].tolist())
    assert np.abs(pos - S["positions"]).max() <= 3e-5 * np.abs(S["positions"]).max()
