cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
tag=r02d
export TMPDIR=/tmp
timeout 120 python tools/bench_contact.py --n 100 --steps 12 > gpurun_out/${tag}_contact_bench.json 2> gpurun_out/${tag}_contact_bench.err
rm -rf /tmp/profc_$tag
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/profc_$tag -o run -- python $GRAFT_REPO_ROOT/tools/bench_contact.py --n 100 --steps 12 > /dev/null 2>&1 )
db=$(find /tmp/profc_$tag -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py $db gpurun_out/${tag}_contact_kernel_stats.md > /dev/null
timeout 100 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench_line.json 2> /dev/null
python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${tag}_contact_bench.json")); print("contact", round(d["ms_per_iter_wall"], 2), d["newton_iterations"], {k: round(v, 2) for k, v in d["split_ms_per_iter"].items()})
except Exception as e:
    print("contact ERR", e)
try:
    d = json.load(open("gpurun_out/${tag}_bench_line.json")); print("bench", round(d["value"], 1), d["roofline"])
except Exception as e:
    print("bench ERR", e)
PY
head -24 gpurun_out/${tag}_contact_kernel_stats.md
