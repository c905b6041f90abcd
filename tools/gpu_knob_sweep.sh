#!/bin/bash
# One-call sweep of the solver's tuning constants on the headline (one run per setting, 150 timed iterations, ~6 s each): the leaf size of the
# dissection had never been swept until the end of round 4 and was worth 3 % -- this is the same pass over the others, at the new leaf size.
#   gpurun --timeout 600 -- 'bash tools/gpu_knob_sweep.sh [outdir] [extra bench flags]'
# Prints it/s, factorisation and sweep ms per setting; winners are to be confirmed with tools/gpu_ab.sh (two interleaved runs each).
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-knobs}
flags=$2
mkdir -p $out
export TMPDIR=/tmp
run() {
  env $1 timeout 60 python bench.py --no-cpu-baseline --no-large --no-contact --steps 150 --warmup 10 $flags 2>/dev/null | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('%-44s %7.1f it/s  factor %.3f  sweeps %.3f' % ('$1', d['value'], d['solver']['factor_ms'], d['solver']['solve_ms']))
except Exception as e:
    print('%-44s failed: %s' % ('$1', e))
" | tee -a $out/sweep.txt
}
run "IPCGPU_UNUSED=1"
for v in 96 128 160 256 320; do run "IPCGPU_MF_XINV_NC=$v"; done
for v in 0 128 256 1024 2048; do run "IPCGPU_MF_SCHUR64_MIN=$v"; done
for v in 120 160 240 280; do run "IPCGPU_MF_NT512_N=$v"; done
for v in 40 60 80; do run "IPCGPU_MF_NT128_N=$v"; done
for v in 2 8 16; do run "IPCGPU_MF_FWD_STRIDE=$v"; done
for v in 768 1280 1536; do run "IPCGPU_MF_BORDER_MAX_NC=$v"; done
for v in 192 208 240; do run "IPCGPU_PATCH_TETS=$v"; done
for v in 11 13 14; do run "IPCGPU_ND_LEAF=$v"; done
run "IPCGPU_UNUSED=2"
