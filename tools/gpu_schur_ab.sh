#!/bin/bash
# A/B of the Schur-complement tile size (IPCGPU_MF_SCHUR64_MIN) in one call: solver parity tests, headline bench and contact bench per variant.
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r03r}
mkdir -p $out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_sharded.py -m gpu -q -x 2>&1 | tail -4 ) > $out/gpu_tests.txt; cat $out/gpu_tests.txt
for v in 1000000000 2048 512 0; do
  IPCGPU_MF_SCHUR64_MIN=$v timeout 300 python bench.py --no-cpu-baseline --no-contact > $out/bench_$v.json 2> /dev/null
  IPCGPU_MF_SCHUR64_MIN=$v timeout 300 python tools/bench_contact.py --n 100 --steps 12 > $out/contact_$v.json 2> /dev/null
  python - <<PY
import json
d = json.load(open("$out/bench_$v.json")); c = json.load(open("$out/contact_$v.json"))
print("SCHUR64_MIN=$v", round(d["value"], 1), "it/s", round(d["ms_per_step"], 4), "ms; solver", round(d["config"]["solver"]["factor_ms"], 3) if "solver" in d["config"] else d.get("solver", {}).get("factor_ms"), "contact", round(c["ms_per_iter_wall"], 3))
PY
done
rm -rf /tmp/profs
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/profs -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-contact > /dev/null 2>&1 )
db=$(find /tmp/profs -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py $db $out/kernel_stats.md > /dev/null && head -9 $out/kernel_stats.md | cut -c1-100
