"""Time-bounded probe of the rocSOLVER csrrf back end (run on the GPU box)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import ipc_amd
from ipc_amd import scene

V, F = scene.make_bar(8, 2, 2, size=(4.0, 0.5, 1.0))
c = ipc_amd.Context(0, solver=1)
c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
c.opt_init(0.025, False)
c.set_positions(scene.twist_state(scene.jitter(V, F), 0.2))
c.set_pattern()
c.assemble_newton(0.025 ** 2, True, with_gradient=False)
a = c.get_a(); ia, ja = c.get_pattern()
t = time.time(); c.analyze_pattern(); print("analyze", time.time() - t, flush=True)
t = time.time(); ok = c.factorize(); print("factorize", ok, time.time() - t, flush=True)
b = np.ones(len(ia) - 1)
t = time.time(); x = c.solve(b); print("solve", time.time() - t, flush=True)
print("resid", np.linalg.norm(c.multiply(x) - b) / np.linalg.norm(b))
c.set_coeff(30, 30, -abs(a[ia[30]]))
t = time.time(); ok = c.factorize(); print("factorize indefinite ->", ok, time.time() - t, flush=True)
x = c.solve(b); print("x finite:", np.isfinite(x).all(), np.abs(x).max())
