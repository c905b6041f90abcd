#!/bin/bash
# Round 5, sixteenth GPU call: one Newton iteration at 375 K nodes launch by launch (rocprofv3 kernel trace + tools/rocprof_timeline.py).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=gpurun_out/r5c16
mkdir -p $out
rm -rf /tmp/prof_c16
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c16 -o run -- python $R/bench.py --no-cpu-baseline --no-contact --no-large --size 433 --steps 6 --warmup 3 > $R/$out/bench433_under_rocprof.json 2> /dev/null )
db=$(find /tmp/prof_c16 -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_timeline.py $db 5 > $out/timeline433.txt 2>&1 && python tools/rocprof_summary.py $db $out/kernel_stats433.md > /dev/null
wc -l $out/timeline433.txt
