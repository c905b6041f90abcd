#!/bin/bash
# Round 5, nineteenth GPU call: the Schur complement in 128 x 128 tiles (k_big_schur128) on the levels with at least IPCGPU_MF_SCHUR128_MIN of them.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r5c19
mkdir -p $out
for s in "IPCGPU_MF_SCHUR128_MIN=0 150" "X=1 433"; do set -- $s; env $1 timeout 200 python tools/check_solver.py $2 2>&1 | tail -1 | sed "s/^/[$1] /"; done | tee $out/residuals.txt
( IPCGPU_MF_SCHUR128_MIN=0 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3 ) | tee $out/gpu_tests.txt
echo "=== mat433"
STEPS=12 bash tools/gpu_ab.sh r5c19/ab433 "--no-contact --size 433" "-" "IPCGPU_MF_SCHUR128_MIN=1000000000" "IPCGPU_MF_SCHUR128_MIN=256" "IPCGPU_MF_SCHUR128_MIN=4096"
echo "=== mat150 (+ contact)"
bash tools/gpu_ab.sh r5c19/ab150 "" "-" "IPCGPU_MF_SCHUR128_MIN=1000000000" "IPCGPU_MF_SCHUR128_MIN=256" "IPCGPU_MF_SCHUR128_MIN=0"
