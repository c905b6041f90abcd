#!/bin/bash
# Round 5, twelfth GPU call: the Schur tiles of a single-workgroup front gathered four at a time (and two children at a time) instead of one after the other
# (default) against the library of the commit before (prev) and groups of two (sg2); the phase probe again; solver tests.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r5c12
mkdir -p $out
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4 ) | tee $out/gpu_tests.txt
IPCGPU_LIB_VARIANT=probe timeout 200 python bench.py --no-cpu-baseline --no-contact --no-large --steps 20 --warmup 5 2>&1 >/dev/null | grep "fused probe" | tee $out/probe_mat150.txt
IPCGPU_LIB_VARIANT=probe timeout 200 python bench.py --no-cpu-baseline --no-contact --no-large --size 433 --steps 4 --warmup 2 2>&1 >/dev/null | grep "fused probe" | tee $out/probe_mat433.txt
echo "=== mat150"
CHECK=1 bash tools/gpu_ab.sh r5c12/ab150 "" "-" "IPCGPU_LIB_VARIANT=prev" "IPCGPU_LIB_VARIANT=sg2"
echo "=== mat433"
STEPS=12 bash tools/gpu_ab.sh r5c12/ab433 "--no-contact --size 433" "-" "IPCGPU_LIB_VARIANT=prev" "IPCGPU_LIB_VARIANT=sg2"
