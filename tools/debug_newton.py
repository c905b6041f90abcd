import sys
import numpy as np
sys.path.insert(0, ".")
import ipc_amd
from ipc_amd import scene
from oracle import orc

def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)

V, F = scene.make_bar(12, 2, 2, size=(5.0, 0.5, 1.0))
left, right = scene.border_verts(V, 0.01)
m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
o = orc.Optimizer(m, dt=0.025, gravity=False, nthreads=4)
o.set_twist(left, right); o.precompute()
c = ipc_amd.Context(0)
c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
c.opt_init(0.025, False); c.set_twist(left, right); c.precompute()
ia, ja = c.get_pattern()
ch = orc.Chol(ia, ja, 4)
for step in range(2):
    o.begin_timestep(); c.begin_timestep()
    print("step", step, "E0", o.state()["E"], c.state()["E"])
    for it in range(12):
        co, cg = o.newton_iter(), c.newton_iter()
        so, sg = o.state(), c.state()
        a = c.get_a()
        ch.factorize(a)
        p_x = ch.solve(-sg["gradient"])  # oracle solver on the GPU's matrix and gradient
        n = len(ia) - 1
        print(it, co, cg, "g", f"{rel(sg['gradient'], so['gradient']):.1e}", "p", f"{rel(sg['searchDir'], so['searchDir']):.1e}",
              "p(gpu vs cpu-solve same A)", f"{rel(sg['searchDir'], p_x):.1e}",
              "E", f"{abs(sg['E']-so['E'])/abs(so['E']):.1e}", "alpha", so["stepSize"], sg["stepSize"], "V", f"{rel(sg['V'], so['V']):.1e}",
              "|p|", np.abs(so["searchDir"]).max(), "tol", so["targetGRes"])
        if co or cg:
            break
    o.end_timestep(); c.end_timestep()
# conditioning
a = c.get_a()
A = np.zeros((n, n))
for r in range(n):
    for k in range(ia[r], ia[r + 1]):
        A[r, ja[k]] = a[k]; A[ja[k], r] = a[k]
w = np.linalg.eigvalsh(A)
print("cond", w.max() / w.min())
