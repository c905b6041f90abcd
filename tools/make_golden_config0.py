"""Golden fixture for BASELINE configs[0], the reference's own hello-world scene input/otherExamples/barTwist_noCollisions.txt
(bar-2523.msh, 886 nodes / 2 523 tets, E = 1e9, nu = 0.4, rho = 1000, dt = 0.025, gravity on, left end fixed, right end turning
270 deg/s about x, self-contact off), on the reference's REAL mesh.

Runs in the build container only (it reads /root/reference/input/tetMeshes/bar-2523.msh through this repo's msh reader, steps
the scene with the CPU oracle) and writes tests/golden/config0_bar2523.npz: the mesh arrays as read, the two Dirichlet
vertex sets, and the oracle's positions / Newton iteration counts after each of the first time steps.  The fixture travels
to the GPU box, where tests/test_gpu_scenes.py compares the HIP stepper with it.

    python tools/make_golden_config0.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from ipc_amd import lib as gl  # noqa: E402
from ipc_amd import scene  # noqa: E402
from oracle import orc  # noqa: E402

MESH = "/root/reference/input/tetMeshes/bar-2523.msh"
STEPS = 3
REL_TOL = 1e-6

V, T, SF = gl.read_tet_mesh(MESH)
left = scene.select_dirichlet(V, SF, (0, 0, 0), (0.01, 1, 1))  # DBC 0 0 0  0.01 1 1  0 0 0  0 0 0
right = scene.select_dirichlet(V, SF, (0.99, 0, 0), (1, 1, 1))  # DBC 0.99 0 0  1 1 1  0 0 0  270 0 0
m = orc.Mesh(V, T, YM=1e9, PR=0.4, density=1000.0)
o = orc.Optimizer(m, dt=0.025, gravity=True, nthreads=8)
orc.opt_add_dirichlet(o, left)
orc.opt_add_dirichlet(o, right, ang_vel_deg=(270, 0, 0))
# The scene starts exactly at rest, where the reference's makePD2d (IglUtils.hpp:138-177) is discontinuous: which way an
# undeformed element's projected Hessian falls depends on round-off, in the reference too.  Individual Newton iterates are
# therefore not comparable between implementations from this start; the minimiser of every incremental potential is, so
# the fixture is taken with a tight Newton tolerance (the scene file's own default is 1e-2).
o.set_rel_tol(REL_TOL)
o.precompute()
pos, iters, energy = [], [], []
for s in range(STEPS):
    iters.append(o.solve_timestep(100))
    st = o.state()
    pos.append(st["V"].copy())
    energy.append(st["E"])
    print(f"step {s}: {iters[-1]} Newton iterations, E = {energy[-1]:.12e}")
out = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "config0_bar2523.npz")
np.savez_compressed(out, V=V, T=T, SF=SF, left=left, right=right, positions=np.array(pos), iters=np.array(iters), energy=np.array(energy), rel_tol=REL_TOL)
print("wrote", os.path.normpath(out), os.path.getsize(out), "bytes;", len(left), "+", len(right), "Dirichlet nodes")
