"""Timing probe (GPU box): assembly + barrier Hessian on the contact benchmark's first constraint set with the projection / the scatter of the
barrier kernel skipped (IPCGPU_HESS_PROBE is read once by the library: one process per variant).  Prints the median ms per call."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipc_amd import lib, scene  # noqa: E402


def main(n=100):
    V, F, nA = scene.make_mat_stack(n, 2, gap=1.2e-3)
    Vs = scene.jitter(V, F, rel=2e-3)
    c = lib.Context(0)
    c.set_mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
    c.set_positions(Vs)
    c.opt_init(0.01, True)
    c.set_surface(scene.surface_tris(F))
    low = np.arange(nA)
    c.set_dbc(low[(np.abs(V[:nA, 0]) > 0.49) | (np.abs(V[:nA, 2]) > 0.49)].astype(np.int32), 1)
    c.enable_self_collision(1e-3)
    vel = np.zeros_like(V)
    vel[nA:, 1] = -0.05
    c.set_velocity(vel)
    c.precompute()
    c.begin_timestep()
    c.newton_iter()
    ts = []
    for rep in range(12):
        t0 = time.perf_counter()
        c.assemble_newton(1e-4, True, with_gradient=False)  # ends with the read-back of the kernel's error word: synchronous
        ts.append(time.perf_counter() - t0)
    print("probe", os.environ.get("IPCGPU_HESS_PROBE", "0"), "lds" if os.environ.get("IPCGPU_HESS_LDS") else "reg", "assemble_newton ms",
          round(1e3 * float(np.median(ts)), 3), "nActive", c.contact_state()["nActive"])


if __name__ == "__main__":
    main()
