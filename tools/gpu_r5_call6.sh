#!/bin/bash
# Round 5, sixth GPU call: one-call sweep of the solver's tuning constants on the round's final kernels (the Schur complement of the single-workgroup fronts moved to
# the matrix cores, which shifts where a front should stop being one) + the step kernel compiled for 4 waves per SIMD (128 registers, 28-44 B of scratch).
#   gpurun --timeout 900 -- 'bash tools/gpu_r5_call6.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r5c6
mkdir -p $out
run() {
  env $1 timeout 60 python bench.py --no-cpu-baseline --no-large --no-contact --steps 150 --warmup 10 $2 2>/dev/null | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('%-44s %7.1f it/s  factor %.3f  sweeps %.3f' % ('$1 $2', d['value'], d['solver']['factor_ms'], d['solver']['solve_ms']))
except Exception as e:
    print('%-44s failed: %s' % ('$1 $2', e))
" | tee -a $out/sweep.txt
}
run "IPCGPU_UNUSED=1"
run "IPCGPU_LIB_VARIANT=occ4"
for v in 48 80 96 128; do run "IPCGPU_MF_FUSED_KB=$v"; done
for v in 120 160 280 100000; do run "IPCGPU_MF_NT512_N=$v"; done
for v in 60 100; do run "IPCGPU_MF_NT128_N=$v"; done
for v in 0 4096; do run "IPCGPU_MF_SCHUR64_MIN=$v"; done
for v in 128 256; do run "IPCGPU_MF_XINV_NC=$v"; done
run "IPCGPU_UNUSED=2"
echo "=== mat433"
run "IPCGPU_UNUSED=1" "--size 433 --steps 12 --warmup 3"
run "IPCGPU_LIB_VARIANT=occ4" "--size 433 --steps 12 --warmup 3"
for v in 96 128; do run "IPCGPU_MF_FUSED_KB=$v" "--size 433 --steps 12 --warmup 3"; done
run "IPCGPU_MF_SCHUR64_MIN=0" "--size 433 --steps 12 --warmup 3"
