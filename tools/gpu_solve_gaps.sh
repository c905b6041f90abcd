#!/bin/bash
# where the backward sweep idles: kernel trace of repeated factorise + solve of the bench matrix (no stepper around it), gaps above 10 us per solve
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
cat > /tmp/solve_loop.py <<'PY'
import sys, numpy as np
import os; sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from ipc_amd import lib, scene
V, F = scene.make_mat(150)
left, right = scene.border_verts(V, 0.01)
c = lib.Context(0)
c.set_mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
c.opt_init(0.04, False)
c.set_twist(left, right, 0.4 * np.pi)
c.precompute(); c.begin_timestep()
for i in range(3): c.newton_iter()
print(c.bench_factor_solve(int(sys.argv[1]) if len(sys.argv) > 1 else 12))
PY
rm -rf /tmp/prof_sl; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_sl -o sl -- python /tmp/solve_loop.py 12 2>/dev/null | tail -1 )
db=$(find /tmp/prof_sl -name "*.db" | head -1)
for w in 6 9 14 20 25; do echo "== solve $w"; python tools/rocprof_timeline.py $db $w | awk '{split($3,a,"="); if (a[2]+0 > 10) print}'; done
python /tmp/solve_loop.py 30 | tail -1
