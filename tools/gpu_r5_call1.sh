#!/bin/bash
# Round 5, first GPU call: (1) bring-up of k_big_step2 (tools/gpu_step2_bringup.sh), (2) the FMA-contraction parity study of the budgeted
# scenes (tools/gpu_nofma_study.py on libipcgpu.so and libipcgpu_nofma.so), (3) a short knob sweep at the round-4 leaf size.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r5_call1.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/nofma gpurun_out/knobs5
echo "=== step2 bring-up"; bash tools/gpu_step2_bringup.sh 2>&1 | tail -25
echo "=== nofma study"
( IPCGPU_LIB_VARIANT=nofma timeout 400 python tools/gpu_nofma_study.py > gpurun_out/nofma/nofma.txt 2> gpurun_out/nofma/nofma.err ) ; grep "^==\|library" gpurun_out/nofma/nofma.txt; tail -2 gpurun_out/nofma/nofma.err
( timeout 400 python tools/gpu_nofma_study.py > gpurun_out/nofma/default.txt 2> gpurun_out/nofma/default.err ) ; grep "^==\|library" gpurun_out/nofma/default.txt; tail -2 gpurun_out/nofma/default.err
echo "=== knob sweep"
out=gpurun_out/knobs5
run() {
  env $1 timeout 60 python bench.py --no-cpu-baseline --no-large --no-contact --steps 150 --warmup 10 2>/dev/null | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('%-44s %7.1f it/s  factor %.3f  sweeps %.3f' % ('$1', d['value'], d['solver']['factor_ms'], d['solver']['solve_ms']))
except Exception as e:
    print('%-44s failed: %s' % ('$1', e))
" | tee -a $out/sweep.txt
}
run "IPCGPU_UNUSED=1"
for v in 128 256 320; do run "IPCGPU_MF_XINV_NC=$v"; done
for v in 0 256 2048; do run "IPCGPU_MF_SCHUR64_MIN=$v"; done
for v in 160 280; do run "IPCGPU_MF_NT512_N=$v"; done
for v in 40 80; do run "IPCGPU_MF_NT128_N=$v"; done
for v in 2 8; do run "IPCGPU_MF_FWD_STRIDE=$v"; done
for v in 768 1536; do run "IPCGPU_MF_BORDER_MAX_NC=$v"; done
for v in 10 14 16; do run "IPCGPU_ND_LEAF=$v"; done
run "IPCGPU_UNUSED=2"
