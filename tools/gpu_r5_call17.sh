#!/bin/bash
# Round 5, seventeenth GPU call: two-level blocking by LEVEL (the step launches of a level move at least IPCGPU_MF_BULK_MIN_MB of own columns) with outer blocks of
# IPCGPU_MF_BULK_BLOCK columns: residuals with everything switched on at the small size, then the sweep at 375 K nodes and the check that 45 K nodes and the contact
# stack do not lose.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r5c17
mkdir -p $out
for s in "IPCGPU_MF_BULK_MIN_MB=0 IPCGPU_MF_BULK_BLOCK=128 150" "IPCGPU_MF_BULK_MIN_MB=0 IPCGPU_MF_BULK_BLOCK=64 150" "X=1 433"; do set -- $s; if [ $# = 3 ]; then e="$1 $2"; n=$3; else e=$1; n=$2; fi; env $e timeout 200 python tools/check_solver.py $n 2>&1 | tail -1 | sed "s/^/[$e] /"; done | tee $out/residuals.txt
( IPCGPU_MF_BULK_MIN_MB=0 IPCGPU_MF_BULK_BLOCK=128 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3 ) | tee $out/gpu_tests.txt
echo "=== mat433"
STEPS=12 bash tools/gpu_ab.sh r5c17/ab433 "--no-contact --size 433" "-" "IPCGPU_MF_BULK_MIN_MB=1e9" "IPCGPU_MF_BULK_MIN_MB=64" "IPCGPU_MF_BULK_MIN_MB=4" "IPCGPU_MF_BULK_BLOCK=128" "IPCGPU_MF_BULK_BLOCK=128 IPCGPU_MF_BULK_MIN_MB=4" "IPCGPU_MF_BULK_BLOCK=512"
echo "=== mat150 (+ contact)"
bash tools/gpu_ab.sh r5c17/ab150 "" "-" "IPCGPU_MF_BULK_MIN_MB=1e9" "IPCGPU_MF_BULK_BLOCK=128"
