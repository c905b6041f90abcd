#!/bin/bash
# Round 5, thirteenth GPU call: the 128-thread variant of the fused kernel on the leaf level (the probe says a leaf is mostly its single-wave pivot inverses: more, narrower
# workgroups per CU?) -- the threshold swept above the widest leaf, which the earlier sweeps (60, 100) never reached.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
echo "=== mat150"
bash tools/gpu_ab.sh r5c13/ab150 "--no-contact" "-" "IPCGPU_MF_NT128_N=120" "IPCGPU_MF_NT128_N=140" "IPCGPU_MF_NT128_N=170" "IPCGPU_MF_NT128_N=250"
echo "=== mat433"
STEPS=12 bash tools/gpu_ab.sh r5c13/ab433 "--no-contact --size 433" "-" "IPCGPU_MF_NT128_N=140" "IPCGPU_MF_NT128_N=250"
