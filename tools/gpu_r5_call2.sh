#!/bin/bash
# Round 5, second GPU call: (1) the whole GPU suite on the pruned library (0 skipped expected: test_gpu_step2.py is gone, the envelope criterion replaces the
# GPU-only scene budgets), (2) A/B of the round's solver changes on one box: XCD-aware Schur tile order, children's sums parked in LDS (3 waves per SIMD in
# k_big_schur64_ea), and three compile-time variants of the step / Schur kernels (ipc_amd/build.py --variant): 3 waves per SIMD, 144 rows per role-B workgroup.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r5_call2.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r5c2
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 ) | tee gpurun_out/r5c2/gpu_tests.txt
S=("-" "IPCGPU_MF_XCD_ORDER=0" "IPCGPU_MF_SCHUR_PARK=0" "IPCGPU_MF_XCD_ORDER=0 IPCGPU_MF_SCHUR_PARK=0" "IPCGPU_LIB_VARIANT=occ3" "IPCGPU_LIB_VARIANT=occ3s" "IPCGPU_LIB_VARIANT=mt3")
echo "=== mat150"
CHECK=1 bash tools/gpu_ab.sh r5c2/ab150 "--no-contact" "${S[@]}"
echo "=== mat433"
STEPS=12 bash tools/gpu_ab.sh r5c2/ab433 "--no-contact --size 433" "${S[@]}"
