#!/bin/bash
# Round 5, ninth GPU call: after the split of mf_numeric.hip and the transport refactoring of bench.py -- the sharded tests (incl. bench.py launched for two ranks),
# the solver checks, and the timeline of one iteration with kernel names (tools/rocprof_timeline.py printed them empty before).
#   gpurun --timeout 900 -- 'bash tools/gpu_r5_call9.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=gpurun_out/r5c9
mkdir -p $out
( timeout 800 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 ) | tee $out/gpu_tests.txt
timeout 120 python tools/check_solver.py 2>&1 | tail -1
rm -rf /tmp/prof_c9
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c9 -o run -- python $R/bench.py --no-cpu-baseline --no-contact --no-large --steps 60 --warmup 10 > $R/$out/bench_line_under_rocprof.json 2> /dev/null )
db=$(find /tmp/prof_c9 -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_timeline.py $db 12 > $out/timeline.txt 2>&1
head -30 $out/timeline.txt
python -c "
import json; d = json.load(open('$out/bench_line_under_rocprof.json')); print(d['value'], d['solver'])"
