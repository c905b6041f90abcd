#!/bin/bash
# Round 5, twenty-first GPU call: the inversion step filter skips the closed-form cubic for elements that provably have no root below the step (k_inversion_step, tMax):
# the whole GPU suite, then the A/B against the library of the commit before.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r5c21
mkdir -p $out
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -5 ) | tee $out/gpu_tests.txt
echo "=== mat150 (+ contact)"
bash tools/gpu_ab.sh r5c21/ab150 "" "-" "IPCGPU_LIB_VARIANT=prev"
echo "=== mat433"
STEPS=12 bash tools/gpu_ab.sh r5c21/ab433 "--no-contact --size 433" "-" "IPCGPU_LIB_VARIANT=prev"
