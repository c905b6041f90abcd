#!/bin/bash
# kernel trace of the bench under the given environment settings -> launch-by-launch timeline of one iteration (tools/iter_timeline.py) + class breakdown
#   tools/gpu_timeline.sh <outdir> "VAR=a OTHER=b" ["<bench flags>"]
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$1
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rm -rf $out/trace
env $2 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python $R/bench.py --no-cpu-baseline --no-large --steps 40 --warmup 10 ${3:---no-contact} > $out/bench_under_trace.json 2> $out/trace.log
cd $R
python tools/iter_breakdown.py $out/trace 12 10 > $out/iter_breakdown.txt 2>&1
python tools/iter_timeline.py $out/trace 14 > $out/iter_timeline.txt 2>&1
head -14 $out/iter_breakdown.txt
find $out/trace -name "*.csv" -size +30M -delete
