import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from ipc_amd import scene, lib
from oracle import orc
orc.build()
def rel(a,b): return np.abs(a-b).max()/max(np.abs(b).max(),1e-300)
for n in (60, 150):
    V, F = scene.make_mat(n)
    left, right = scene.border_verts(V, 0.01)
    SF = scene.surface_tris(F)
    out = {}
    for name, ip, surf in (("gpu_off", 0, 0), ("gpu_surf", 0, 1), ("gpu_ip", 1, 1)):
        c = lib.Context(0)
        c.set_mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
        c.opt_init(0.04, False)
        if surf: c.set_surface(SF)
        c.set_twist(left, right, 0.4*np.pi)
        if ip: c.enable_self_collision(1e-3)
        c.precompute(); c.begin_timestep()
        d = c.dbc_state()
        c.newton_iter(); s = c.state(); s["dbc"] = d
        ia, ja = c.get_pattern(); s["nnz"] = len(ja)
        out[name] = s
        c.close()
    m = orc.Mesh(V, F, YM=2e4, PR=0.4, density=1000.0); m.set_surface(SF)
    o = orc.Optimizer(m, dt=0.04, gravity=False, nthreads=16)
    o.set_twist(left, right, 0.4*np.pi); orc.opt_enable_self_collision(o, 1e-3)
    o.precompute(); o.begin_timestep(); d = orc.opt_dbc_state(o); o.newton_iter(); so = o.state()
    print("n", n, "oracle dbc", d, "E", so["E"])
    for k, s in out.items():
        print("  ", k, "dbc", s["dbc"], "nnz", s["nnz"], "E", s["E"], "step", s["stepSize"], "dP vs oracle", rel(s["searchDir"], so["searchDir"]), "dG", rel(s["gradient"], so["gradient"]), "dV", rel(s["V"], so["V"]),
              "dP vs gpu_off", rel(s["searchDir"], out["gpu_off"]["searchDir"]))
