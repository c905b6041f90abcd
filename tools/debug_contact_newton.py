"""Side-by-side trace of the contact-aware stepper (GPU vs oracle); prints where the two first part ways.
usage: debug_contact_newton.py [jitter] [steps] [n] [speed] [dt]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import orc  # noqa: E402
import ipc_amd.lib as gpu_lib  # noqa: E402
from test_gpu_contact import drop_scene, relerr  # noqa: E402

arg = lambda i, d: type(d)(sys.argv[i]) if len(sys.argv) > i else d  # noqa: E731
jit, steps, n, speed, dt = arg(1, 1e-2), arg(2, 3), arg(3, 2), arg(4, -1.5), arg(5, 0.01)
m, o, c, nA = drop_scene(orc, gpu_lib, n=n, speed=speed, dt=dt, jitter=jit)
o.precompute()
c.precompute()
for step in range(steps):
    o.begin_timestep()
    c.begin_timestep()
    so, sg = o.state(), c.state()
    print(f"step {step}: begin E {so['E']:.12e} {sg['E']:.12e} kappa {so['kappa']:.6e} {sg['kappa']:.6e}")
    if step == 0:
        cs = orc.Contacts()
        cs.build(m, so["dHat"])
        pairs = cs.connectivity(m)
        ia, ja = m.pattern(extra_edges=pairs)
        a_e = m.assemble_hessian(len(ja), dt * dt, True)
        a_c = cs.hessian(m, len(ja), so["dHat"], so["kappa"], True)
        c.assemble_newton(dt * dt, True, with_gradient=False)
        a_g = c.get_a()
        ia_g, ja_g = c.get_pattern()
        print("   pattern equal", np.array_equal(ia, ia_g), np.array_equal(ja, ja_g), "nnz", len(ja), len(ja_g))
        if len(a_g) == len(a_e):
            d = np.abs(a_g - (a_e + a_c))
            print("   full a relerr", d.max() / np.abs(a_e + a_c).max())
    for it in range(60):
        co, cg = o.newton_iter(), c.newton_iter()
        so, sg = o.state(), c.state()
        ko, kg = orc.opt_contact_state(o), c.contact_state()
        print(f"  it {it}: conv {co} {cg} | E {so['E']:.10e} {sg['E']:.10e} | aF {so['alphaFeasible']:.6e} {sg['alphaFeasible']:.6e} "
              f"| a {so['stepSize']:.4e} {sg['stepSize']:.4e} | kappa {so['kappa']:.4e} {sg['kappa']:.4e} | nA {len(ko['active'])} {kg['nActive']} "
              f"| fullCCD {ko['n_full_ccd']} {kg['nFullCCD']} | g {relerr(sg['gradient'], so['gradient']):.2e} "
              f"p {relerr(sg['searchDir'], so['searchDir']):.2e} V {relerr(sg['V'], so['V']):.2e}")
        if co or cg:
            break
    o.end_timestep()
    c.end_timestep()
