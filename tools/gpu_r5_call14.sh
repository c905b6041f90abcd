#!/bin/bash
# Round 5, fourteenth GPU call: two-level blocking of the wide fronts (outer blocks of 256 own columns, one bulk update per block: k_big_bulk).
# Correctness where it is switched on (residuals at three sizes, the solver tests with the threshold lowered so that mat150's root takes the path), then the A/B against
# IPCGPU_MF_BULK_MIN_NC=1000000 (every front updates all its own columns step by step, as before).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r5c14
mkdir -p $out
for s in "IPCGPU_MF_BULK_MIN_NC=512 150" "IPCGPU_MF_BULK_MIN_NC=1000000 300" "X=1 300" "X=1 433"; do set -- $s; env $1 timeout 200 python tools/check_solver.py $2 2>&1 | tail -1 | sed "s/^/[$1] /"; done | tee $out/residuals.txt
( IPCGPU_MF_BULK_MIN_NC=512 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4 ) | tee $out/gpu_tests.txt
echo "=== mat433"
STEPS=12 bash tools/gpu_ab.sh r5c14/ab433 "--no-contact --size 433" "-" "IPCGPU_MF_BULK_MIN_NC=1000000" "IPCGPU_MF_BULK_MIN_NC=512"
echo "=== mat150 (+ contact sub-records: the stack's root has 1 200 columns)"
bash tools/gpu_ab.sh r5c14/ab150 "" "-" "IPCGPU_MF_BULK_MIN_NC=1000000" "IPCGPU_MF_BULK_MIN_NC=512"
