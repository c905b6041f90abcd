#!/bin/bash
# one gpurun call: the assembly pass alone (tools/probe_assembly.py) under a few plan / probe settings, then a rocprofv3 kernel trace of it
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r03asm}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
for n in 150 433; do
  for cfg in "X=0" "IPCGPU_ASM_PROBE=1" "IPCGPU_ASM_PROBE=2" $EXTRA_CFGS; do
    env IPCGPU_PLAN_STATS=1 $cfg timeout 300 python tools/probe_assembly.py $n 2>&1 | grep -v amdgpu.ids | sed "s/^/[$cfg] /"
  done
done > gpurun_out/$tag/probe.txt 2>&1
cat gpurun_out/$tag/probe.txt
( cd /tmp && rm -rf /tmp/prof_$tag && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o run -- python $GRAFT_REPO_ROOT/tools/probe_assembly.py 150 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/$tag/prof.err )
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocprof_summary.py $db gpurun_out/$tag/kernel_stats.md > /dev/null; head -20 gpurun_out/$tag/kernel_stats.md; else tail -5 gpurun_out/$tag/prof.err; fi
