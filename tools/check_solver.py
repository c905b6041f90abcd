"""Quick hardware check of the sparse solver at the benched size: assemble the mat150 Newton matrix, factorise, solve two right-hand
sides, print the relative residuals (computed on the GPU with the CSR product) and the not-PD behaviour.  A few seconds; the
oracle comparison lives in tests/test_gpu_fullsize.py.  usage: python tools/check_solver.py [size]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipc_amd import lib, scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
V, F = scene.make_mat(n)
Vt = scene.twist_state(scene.jitter(V, F), 0.5)
left, right = scene.border_verts(V, 0.01)
c = lib.Context(0)
c.set_mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
c.opt_init(0.04, False)
c.set_dbc(np.concatenate([left, right]), 2)
c.set_positions(Vt)
c.set_pattern()
c.assemble_newton(0.04 ** 2, True, with_gradient=False)
c.analyze_pattern()
ok = c.factorize()
rows, nnz = c.get_dims()
worst = 0.0
for seed in (1, 2):
    b = np.random.default_rng(seed).normal(size=rows)
    x = c.solve(b)
    r = np.linalg.norm(c.multiply(x) - b) / np.linalg.norm(b)
    worst = max(worst, r)
ok2 = c.factorize()
b = np.random.default_rng(3).normal(size=rows)
r3 = np.linalg.norm(c.multiply(c.solve(b)) - b) / np.linalg.norm(b)
a = c.get_a()
ia, ja = c.get_pattern()
k = ia[3 * (rows // 6)]
c.set_coeff(3 * (rows // 6), 3 * (rows // 6), -abs(a[k]))
bad = c.factorize()
print(f"check_solver mat{n}: factorize={ok}/{ok2} residuals {worst:.2e} {r3:.2e} notPD_detected={not bad}")
assert ok and ok2 and worst < 1e-10 and r3 < 1e-10 and not bad
c.close()
