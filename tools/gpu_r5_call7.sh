#!/bin/bash
# Round 5, seventh GPU call: v_permlane16_swap_b32 instead of ds_bpermute for the partner row of a 2 x 2 pivot (ipc_amd/libipcgpu_pl16.so, -DMF_XOR16_PERMLANE):
# the instruction's semantics first (tools/probe_permlane.hip), then residuals, then the A/B; and the widest front whose inverse grows by bordering, on the contact bench.
#   gpurun --timeout 900 -- 'bash tools/gpu_r5_call7.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r5c7
mkdir -p $out
tools/_build/probe_permlane | tee $out/probe_permlane.txt
IPCGPU_LIB_VARIANT=pl16 timeout 120 python tools/check_solver.py 2>&1 | tail -1
echo "=== mat150 (+ contact sub-records)"
bash tools/gpu_ab.sh r5c7/ab150 "" "-" "IPCGPU_LIB_VARIANT=pl16"
echo "=== mat433"
STEPS=12 bash tools/gpu_ab.sh r5c7/ab433 "--no-contact --size 433" "-" "IPCGPU_LIB_VARIANT=pl16"
echo "=== contact bench"
for s in "X=0" "IPCGPU_LIB_VARIANT=pl16" "IPCGPU_MF_BORDER_MAX_NC=512" "IPCGPU_MF_BORDER_MAX_NC=768" "IPCGPU_MF_BORDER_MAX_NC=1536" "X=1" "IPCGPU_LIB_VARIANT=pl16"; do env $s timeout 300 python tools/bench_contact.py --n 100 --steps 12 2>/dev/null | python -c "
import sys, json
d = json.load(sys.stdin); print('%-32s' % '$s', round(d['ms_per_iter_wall'], 3), d['newton_iterations'], {k[:24]: round(v, 2) for k, v in d['split_ms_per_iter'].items()})"; done | tee $out/contact_ab.txt | cut -c1-260
