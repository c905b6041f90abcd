#!/bin/bash
# Collects the round's measurement artefacts on the GPU box into gpurun_out/ (copied into profiles/ afterwards):
# bench line (with the CPU baseline), rocprofv3 kernel table + one factor/solve timeline of the same command, the 1.12 M-tet size,
# the contact benchmark with its kernel table.   usage: bash tools/gpu_profiles.sh TAG
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
tag=${1:-r02}
export TMPDIR=/tmp
timeout 600 python bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
rm -rf /tmp/prof_$tag
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_line_under_rocprof.json 2> /dev/null )
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py $db gpurun_out/${tag}_kernel_stats.md > /dev/null && python tools/rocprof_timeline.py $db 12 > gpurun_out/${tag}_timeline.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --size 433 --steps 10 --warmup 3 > gpurun_out/${tag}_bench_mat433.json 2> /dev/null
timeout 300 python tools/bench_contact.py --n 100 --steps 12 > gpurun_out/${tag}_contact_bench.json 2> /dev/null
rm -rf /tmp/profc_$tag
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/profc_$tag -o run -- python $GRAFT_REPO_ROOT/tools/bench_contact.py --n 100 --steps 12 > /dev/null 2>&1 )
db=$(find /tmp/profc_$tag -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py $db gpurun_out/${tag}_contact_kernel_stats.md > /dev/null
python - <<PY
import json
for f in ("${tag}_bench_line", "${tag}_bench_mat433"):
    try:
        d = json.load(open("gpurun_out/%s.json" % f))
        print(f, round(d["value"], 1), {k: round(v, 3) for k, v in d["split_ms_per_iter"].items()}, "asm frac", round(d["roofline"]["frac"], 4),
              [round(r["frac"], 4) for r in d.get("roofline_solver", [])], d.get("cpu_baseline", {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
try:
    d = json.load(open("gpurun_out/${tag}_contact_bench.json")); print("contact", round(d["ms_per_iter_wall"], 2), {k: round(v, 2) for k, v in d["split_ms_per_iter"].items()})
except Exception as e:
    print("contact ERR", e)
PY
head -14 gpurun_out/${tag}_contact_kernel_stats.md
