cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
tag=r02f
export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
( cd /tmp && timeout 70 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_line_under_rocprof.json 2> /dev/null )
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py $db gpurun_out/${tag}_kernel_stats.md > /dev/null
head -12 gpurun_out/${tag}_kernel_stats.md
head -c 400 gpurun_out/${tag}_bench_line_under_rocprof.json
