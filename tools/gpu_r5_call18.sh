#!/bin/bash
# Round 5, eighteenth GPU call: the 64 x 64 Schur tile with its operand chunks requested two products ahead (three register sets, ipc_amd/libipcgpu_pf2.so) instead of one.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
IPCGPU_LIB_VARIANT=pf2 timeout 200 python tools/check_solver.py 150 2>&1 | tail -1
IPCGPU_LIB_VARIANT=pf2 timeout 200 python tools/check_solver.py 433 2>&1 | tail -1
echo "=== mat433"
STEPS=12 bash tools/gpu_ab.sh r5c18/ab433 "--no-contact --size 433" "-" "IPCGPU_LIB_VARIANT=pf2"
echo "=== mat150 (+ contact)"
bash tools/gpu_ab.sh r5c18/ab150 "" "-" "IPCGPU_LIB_VARIANT=pf2"
