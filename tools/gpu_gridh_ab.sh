#!/bin/bash
# cell size of the narrow phase's grid (library variants -DGRID_H_SCALE=1.5 / 2.0 against 1.0 mean edge lengths): contact bench under rocprofv3 (kernel times) and plain, matTwist early
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in "" h15 h20; do
  rm -rf /tmp/prof_gh$v
  ( cd /tmp && IPCGPU_LIB_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_gh$v -o run -- python $GRAFT_REPO_ROOT/tools/bench_contact.py --n 100 --layers 2 --steps 12 --max-iter 12 > /dev/null 2>&1 )
  db=$(find /tmp/prof_gh$v -name "*.db" | head -1)
  python tools/rocprof_summary.py $db /tmp/gh$v.md > /dev/null
  echo "variant [$v]: $(grep 'k_grid_insert_both\|k_narrow_pt\|k_narrow_ee_cells' /tmp/gh$v.md | cut -c1-64 | tr '\n' ' ')"
  IPCGPU_LIB_VARIANT=$v timeout 300 python tools/bench_contact.py --n 100 --layers 2 --steps 12 --max-iter 12 > /tmp/ghc$v.json 2>/dev/null
  IPCGPU_LIB_VARIANT=$v timeout 300 python tools/bench_mat_twist.py > /tmp/ght$v.json 2>/dev/null
  python - /tmp/ghc$v.json /tmp/ght$v.json <<'PY'
import json, sys
c = json.load(open(sys.argv[1])); t = json.load(open(sys.argv[2]))
print("   contact %.3f ms/iter (constraint_sets %.3f) | twist early %.3f (constraint_sets %.3f) wrapped %.3f" % (c["ms_per_iter_wall"], c["split_ms_per_iter"]["constraint_sets"], t["early"]["ms_per_iter"], t["early"]["split_ms_per_iter"]["constraint_sets"], t["wrapped"]["ms_per_iter"]))
PY
done
