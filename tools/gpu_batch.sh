#!/bin/bash
# one gpurun call: selected GPU tests, bench, solver knob sweeps (outputs under gpurun_out/)
# usage: gpu_batch.sh TAG "pytest args" [sweep]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
tag=${1:-r02x}
sel=${2:-tests -m gpu}
( timeout 1200 python -m pytest $sel -x -q 2>&1 | tail -25 ) > gpurun_out/${tag}_tests.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
if [ -n "$3" ]; then
  for kv in $3; do
    env $kv timeout 200 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/${tag}_bench_${kv//=/_}.json 2>/dev/null
  done
fi
cat gpurun_out/${tag}_tests.log
tail -3 gpurun_out/${tag}_bench.err
for f in gpurun_out/${tag}_bench*.json; do echo $f; python - <<PY
import json
try:
    d=json.load(open("$f")); print(round(d["value"],1), {k: round(v,3) for k,v in d["split_ms_per_iter"].items()}, round(d["solver"]["factor_ms"],3), round(d["solver"]["solve_ms"],3), d["solver"]["fronts"], d["solver"]["levels"])
except Exception as e: print("ERR", e)
PY
done
