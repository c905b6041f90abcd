#!/bin/bash
# kernel trace of the bench command -> where an iteration's wall time goes (tools/iter_breakdown.py).  Outputs: gpurun_out/$1/
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-r03w2}
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python $R/bench.py --no-cpu-baseline --no-contact > $out/bench_under_trace.json 2> $out/trace.log
cd $R
python tools/iter_breakdown.py $out/trace 12 10 | tee $out/iter_breakdown.txt
IPCGPU_MF_NO_FWD_OVERLAP=1 timeout 600 python bench.py --no-cpu-baseline --no-contact > $out/bench_no_fwd_overlap.json 2>> $out/trace.log
timeout 600 python bench.py --no-cpu-baseline --no-contact > $out/bench_default.json 2>> $out/trace.log
timeout 600 python bench.py --no-cpu-baseline --no-contact > $out/bench_no_graph.json 2>> $out/trace.log
python - <<PY
import json
for n in ("bench_under_trace", "bench_default", "bench_no_fwd_overlap", "bench_no_graph"):
    try:
        d = json.load(open("$out/" + n + ".json")); print(n, round(d["value"], 1), "it/s", d["solver"]["factor_ms"], d["solver"]["solve_ms"])
    except Exception as e:
        print(n, "failed", e)
PY
find $out/trace -name "*.csv" -size +20M -delete
