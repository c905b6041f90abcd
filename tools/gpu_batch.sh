#!/bin/bash
# one gpurun call: selected GPU tests, bench, optional env sweeps and a rocprofv3 kernel trace (outputs under gpurun_out/)
# usage: TESTS="pytest args" SWEEP="K=V K=V" PROF=1 bash tools/gpu_batch.sh TAG
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
tag=${1:-r02x}
export TMPDIR=/tmp
if [ -n "$TESTS" ]; then
  ( eval "timeout 1500 python -m pytest $TESTS -x -q" 2>&1 | tail -25 ) > gpurun_out/${tag}_tests.log
  cat gpurun_out/${tag}_tests.log
fi
timeout 300 python bench.py --no-cpu-baseline $BENCHARGS > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
for kv in $SWEEP; do
  env $kv timeout 200 python bench.py --no-cpu-baseline --steps 20 $BENCHARGS > gpurun_out/${tag}_bench_${kv//=/_}.json 2>/dev/null
done
if [ -n "$PROF" ]; then
  rm -rf /tmp/prof_$tag
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 $BENCHARGS > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.err )
  db=$(find /tmp/prof_$tag -name "*.db" | head -1)
  if [ -n "$db" ]; then
    python tools/rocprof_summary.py $db gpurun_out/${tag}_kernel_stats.md > /dev/null
    python tools/rocprof_timeline.py $db 12 > gpurun_out/${tag}_timeline.txt 2>&1
  else
    echo "no rocprof db" ; tail -5 gpurun_out/${tag}_prof.err
  fi
fi
tail -3 gpurun_out/${tag}_bench.err
for f in gpurun_out/${tag}_bench*.json; do echo $f; python - <<PY
import json
try:
    d=json.load(open("$f")); print(round(d["value"],1), {k: round(v,3) for k,v in d["split_ms_per_iter"].items()}, round(d["solver"]["factor_ms"],3), round(d["solver"]["solve_ms"],3), d["solver"]["fronts"], d["solver"]["levels"])
except Exception as e: print("ERR", e)
PY
done
