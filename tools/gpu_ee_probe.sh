#!/bin/bash
# where k_narrow_ee_cells spends its time: library variants built with -DEE_PROBE=1 (pairs found and queued, not typed) / 2 (records fetched, no pair loop)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in "" ee1 ee2; do
  rm -rf /tmp/prof_ee$v
  ( cd /tmp && IPCGPU_LIB_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ee$v -o run -- python $GRAFT_REPO_ROOT/tools/bench_contact.py --n 100 --layers 2 --steps 3 --max-iter 6 > /dev/null 2>&1 )
  db=$(find /tmp/prof_ee$v -name "*.db" | head -1)
  python tools/rocprof_summary.py $db /tmp/ee$v.md > /dev/null
  echo "variant [$v]: $(grep 'k_narrow_ee_cells\|k_grid_insert_both\|k_narrow_pt' /tmp/ee$v.md | tr '\n' ' ')"
done
