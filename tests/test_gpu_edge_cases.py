"""Edge cases of the C ABI on the GPU: smallest meshes, empty sets, argument errors that must come back as status codes with a
message (no exception crosses the boundary, nothing falls back to the CPU)."""
import numpy as np
import pytest

from ipc_amd import scene

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_single_tetrahedron_steps_like_the_oracle(orc, gpu_lib):
    V = np.array([[0, 0, 0], [1.0, 0, 0], [0, 1.1, 0], [0.1, 0.2, 0.9]])
    F = np.array([[0, 1, 2, 3]], dtype=np.int32)
    X = V + 0.05 * np.random.default_rng(0).normal(size=V.shape)
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_V(X)
    o = orc.Optimizer(m, dt=0.01, gravity=True, nthreads=1)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    c.set_positions(X)
    c.opt_init(0.01, True)
    o.precompute()
    c.precompute()
    for _ in range(3):
        assert o.solve_timestep(50) == c.solve_timestep(50)
        assert relerr(c.state()["V"], o.state()["V"]) < 1e-9
    c.close()


def test_contact_machinery_with_nothing_in_contact(orc, gpu_lib):
    """Two blocks far apart: empty constraint sets, unit CCD bound, no pattern change, and the stepper still matches."""
    Va, Fa = scene.make_box(2, 1, 1, size=(1.0, 0.5, 0.5), origin=(0, 0, 0))
    Vb, Fb = scene.make_box(1, 1, 1, size=(0.5, 0.5, 0.5), origin=(0, 3.0, 0))
    V = np.vstack([Va, Vb])
    F = np.vstack([Fa, Fb + Va.shape[0]]).astype(np.int32)
    Vs = scene.jitter(V, F, rel=1e-2)
    SF = scene.surface_tris(F)
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_surface(SF)
    m.set_V(Vs)
    o = orc.Optimizer(m, dt=0.01, gravity=False, nthreads=2)
    orc.opt_enable_self_collision(o, 1e-3)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    c.set_positions(Vs)
    c.opt_init(0.01, False)
    c.set_surface(SF)
    c.enable_self_collision(1e-3)
    o.precompute()
    c.precompute()
    for _ in range(2):
        assert o.solve_timestep(50) == c.solve_timestep(50)
        cs = c.contact_state()
        assert cs["nActive"] == 0 and cs["nPara"] == 0 and cs["nPatternChanges"] == 0
        assert relerr(c.state()["V"], o.state()["V"]) < 1e-9
    assert not c.is_intersected()
    c.close()


def test_errors_come_back_as_status_codes(gpu_lib, tmp_path):
    V, F = scene.make_box(1, 1, 1)
    c = gpu_lib.Context(0)
    with pytest.raises(gpu_lib.IpcGpuError):
        c.opt_init(0.01, False)  # no mesh yet
    c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    with pytest.raises(gpu_lib.IpcGpuError):
        c._chk(c._L.ipcgpu_set_energy_type(c.h, 7))  # neither NH nor FCR
    c.opt_init(0.01, False)
    with pytest.raises(gpu_lib.IpcGpuError):
        c.add_dirichlet(np.array([V.shape[0] + 3], dtype=np.int32))  # vertex id out of range
    with pytest.raises(gpu_lib.IpcGpuError):
        c.load_status(tmp_path / "nope")
    bad = tmp_path / "status_bad"
    open(bad, "w").write("timestep 1\n\nposition 999 3\n0 0 0\n")
    with pytest.raises(gpu_lib.IpcGpuError):
        c.load_status(bad)  # more rows than the mesh has
    with pytest.raises(gpu_lib.IpcGpuError):
        c.begin_timestep()  # before precompute
    c.close()


def test_malformed_mmcvid_tuples_are_refused(gpu_lib):
    """ipcgpu_contact_evaluate / ipcgpu_contact_jt_multiply (SelfCollisionHandler.cpp:37-148 on caller-supplied tuples): every node id a kernel would read is
    range-checked per stencil kind, exactly as the device decodes the tuple -- a negative id behind an edge-edge head or a node id past the mesh must come back
    as an argument error, never reach the device."""
    import ctypes as C
    V, F = scene.make_box(2, 2, 2)
    nV = V.shape[0]
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    c.opt_init(0.01, False)
    c.set_surface(scene.surface_tris(F))
    L = c._L
    L.ipcgpu_contact_evaluate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.ipcgpu_contact_jt_multiply.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]

    def both(t):
        t = np.ascontiguousarray(t, dtype=np.int32).reshape(-1, 4)
        val, inp, out = np.zeros(len(t)), np.ones(len(t)), np.zeros(3 * nV)
        return (L.ipcgpu_contact_evaluate(c.h, len(t), t.ctypes.data, val.ctypes.data),
                L.ipcgpu_contact_jt_multiply(c.h, len(t), t.ctypes.data, inp.ctypes.data, 1.0, out.ctypes.data), val, out)
    good = [[0, 1, 22, 26], [-1, 26, -1, -2], [-1, 25, 26, -3], [-1, 22, 25, 26]]  # EE, PP x2, PE x3, PT (non-parallel edges, a proper triangle: the 0 / 0 of the degenerate ones is NaN here as in the reference)
    r0, r1, val, out = both(good)
    assert r0 == 0 and r1 == 0 and np.all(val > 0) and np.abs(out).max() > 0
    for bad in ([5, 6, 7, -3], [5, -2, 7, 8], [5, 6, -1, 8], [-4, -2, 1, 2], [-1, nV, -1, -1], [-nV - 1, 2, -1, -1], [-1, 2, nV, -1], [-1, 2, 3, nV], [0, 1, 2, nV]):
        r0, r1, _, out = both(good + [bad])
        assert r0 != 0 and r1 != 0, bad
        assert np.all(out == 0.0)  # refused before anything was added
    c.close()


def test_a_setter_between_newton_iterations_discards_the_assembly_enqueued_ahead(orc, gpu_lib):
    """On the contact-free path the stepper enqueues the next iteration's assembly behind an accepted trial (HipOptimizer::speculativeAssembly).
    A caller that changes the constraints between two ipcgpu_opt_newton_iter calls must get an assembly of the NEW state: same iterates as the
    oracle driven the same way (a stale assembly would project the Dirichlet rows of the old node set)."""
    V, F = scene.make_mat(10)
    X = scene.jitter(V, F, rel=5e-2)
    left = np.where(V[:, 0] < -0.49)[0].astype(np.int32)
    right = np.where(V[:, 0] > 0.49)[0].astype(np.int32)
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_V(X)
    m.set_dbc(left, 1)
    o = orc.Optimizer(m, dt=0.01, gravity=True, nthreads=1)
    c = gpu_lib.Context(0)
    c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    c.set_positions(X)
    c.set_dbc(left, 1)
    c.opt_init(0.01, True)
    o.precompute()
    c.precompute()
    o.begin_timestep()
    c.begin_timestep()
    for _ in range(2):
        assert bool(o.newton_iter()) == bool(c.newton_iter())
    m.clear_dbc()
    m.set_dbc(right, 1)
    c.clear_dbc()
    c.set_dbc(right, 1)
    for _ in range(3):
        assert bool(o.newton_iter()) == bool(c.newton_iter())
        assert relerr(c.state()["V"], o.state()["V"]) < 1e-9
    c.close()
