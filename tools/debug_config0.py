import os, sys
import numpy as np
sys.path.insert(0, ".")
import ipc_amd as gpu_lib
from oracle import orc
g = np.load("tests/golden/config0_bar2523.npz")
V, T = g["V"], g["T"]
m = orc.Mesh(V, T, YM=1e9, PR=0.4, density=1000.0)
o = orc.Optimizer(m, dt=0.025, gravity=True, nthreads=4)
orc.opt_add_dirichlet(o, g["left"])
orc.opt_add_dirichlet(o, g["right"], ang_vel_deg=(270, 0, 0))
c = gpu_lib.Context(0)
c.set_mesh(V, T, YM=1e9, PR=0.4, density=1000.0)
c.opt_init(0.025, True)
c.add_dirichlet(g["left"])
c.add_dirichlet(g["right"], ang_vel_deg=(270, 0, 0))
o.precompute(); c.precompute()
o.begin_timestep(); c.begin_timestep()
xk = o.state()["V"].copy()
so, sg = o.state(), c.state()
print("after begin: dV", np.abs(sg["V"] - so["V"]).max(), "E", so["E"], sg["E"])
print("moved o", np.abs(so["V"] - V).max(), "moved g", np.abs(sg["V"] - V).max())
co, cg = o.newton_iter(), c.newton_iter()
so, sg = o.state(), c.state()
print("it0: dV", np.abs(sg["V"] - so["V"]).max(), "E", so["E"], sg["E"], "step", so["stepSize"], sg["stepSize"])
print("grad diff", np.abs(sg["gradient"] - so["gradient"]).max(), np.abs(so["gradient"]).max())
d = np.abs(sg["gradient"] - so["gradient"]).reshape(-1, 3).max(1)
bad = np.argsort(-d)[:8]
print("worst grad nodes", bad, d[bad], "in left", np.isin(bad, g["left"]), "in right", np.isin(bad, g["right"]))
print("searchDir diff", np.abs(sg["searchDir"] - so["searchDir"]).max(), np.abs(so["searchDir"]).max())
# --- where do the matrices differ?
a_g = c.get_a()
ia, ja = c.get_pattern()
Vn = so["V"].copy()
m.set_V(xk)
a_o = m.assemble_hessian(len(ja), 0.025 ** 2, projectDBC=True)
d = np.abs(a_g - a_o)
print("matrix diff", d.max(), "of", np.abs(a_o).max(), "entries off by > 1e-6 rel:", (d > 1e-6 * np.abs(a_o).max()).sum())
rows = np.repeat(np.arange(len(ia) - 1), np.diff(ia))
badk = np.nonzero(d > 1e-6 * np.abs(a_o).max())[0]
nodes = np.unique(np.concatenate([rows[badk] // 3, ja[badk] // 3]))
print("nodes involved", nodes[:20], "in right:", np.isin(nodes, g["right"]).sum(), "in left:", np.isin(nodes, g["left"]).sum(), "of", len(nodes))
tets = np.nonzero(np.isin(T, nodes).sum(1) >= 2)[0][:6]
for t in tets:
    X = xk[T[t]]; R0 = V[T[t]]
    Dm = (R0[1:] - R0[0]).T; Ds = (X[1:] - X[0]).T
    F = Ds @ np.linalg.inv(Dm)
    print("tet", t, "sigma", np.linalg.svd(F)[1], "det", np.linalg.det(F))
    Ho = m.elastic_hessian_elem(int(t), 0.025 ** 2, True)
    print("   elem H eig min/max", np.linalg.eigvalsh(Ho)[[0, -1]])
