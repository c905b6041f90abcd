#!/bin/bash
# the reference CCD sweep's search cells: m x m x m of the reference's voxels per cell (acceptance stays on the fine voxel indices: the same pairs) -- library variants -DREF_GRID_MIN_M=2 / 3
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in "" m2 m3; do
  rm -rf /tmp/prof_rg$v
  ( cd /tmp && IPCGPU_LIB_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_rg$v -o run -- python $GRAFT_REPO_ROOT/tools/bench_contact.py --n 100 --layers 2 --steps 12 --max-iter 12 > /tmp/rg$v.json 2>/dev/null )
  db=$(find /tmp/prof_rg$v -name "*.db" | head -1)
  python tools/rocprof_summary.py $db /tmp/rg$v.md > /dev/null
  echo "variant [$v]: $(grep 'k_ref_insert\|k_ref_sweep' /tmp/rg$v.md | cut -c1-70 | tr '\n' ' ')"
  python - /tmp/rg$v.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("   under rocprof: %.3f ms/iter, timestep %.3f step_bounds %.3f" % (d["ms_per_iter_wall"], d["split_ms_per_iter"]["timestep"], d["split_ms_per_iter"]["step_bounds(inversion+CCD+CFL)"]), [c["nFullCCD"] for c in d["contact_state_per_step"]][-1])
PY
done
for v in "" m2 m3; do
  IPCGPU_LIB_VARIANT=$v timeout 300 python tools/bench_mat_twist.py > /tmp/tw$v.json 2>/dev/null
  python - /tmp/tw$v.json "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
e = d["early"]
print("twist early [%s]: %.3f ms/iter timestep %.3f step_bounds %.3f" % (sys.argv[2], e["ms_per_iter"], e["split_ms_per_iter"]["timestep"], e["split_ms_per_iter"]["step_bounds(inversion+CCD+CFL)"]))
PY
done
