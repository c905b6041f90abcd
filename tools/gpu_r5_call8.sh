#!/bin/bash
# Round 5, eighth GPU call: the entries' destinations and the per-front / per-tile entry lists are built on the device (k_entry_dst, k_scan_exclusive,
# k_entry_scatter in MfNumeric::setup) instead of on the host.  The GPU suite (with the new test that pins the kernel on the host function), then the A/B against
# the library built from the commit before (ipc_amd/libipcgpu_prev.so): headline and mat433 must not move, the contact bench's symbolic share should fall.
#   gpurun --timeout 1200 -- 'bash tools/gpu_r5_call8.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r5c8
mkdir -p $out
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 ) | tee $out/gpu_tests.txt
echo "=== mat150"
CHECK=1 bash tools/gpu_ab.sh r5c8/ab150 "" "-" "IPCGPU_LIB_VARIANT=prev"
echo "=== mat433"
STEPS=12 bash tools/gpu_ab.sh r5c8/ab433 "--no-contact --size 433" "-" "IPCGPU_LIB_VARIANT=prev"
echo "=== contact bench"
for s in "X=0" "IPCGPU_LIB_VARIANT=prev" "X=1" "IPCGPU_LIB_VARIANT=prev"; do env $s timeout 300 python tools/bench_contact.py --n 100 --steps 12 2>/dev/null | python -c "
import sys, json
d = json.load(sys.stdin); print('%-32s' % '$s', round(d['ms_per_iter_wall'], 3), d['newton_iterations'], {k[:24]: round(v, 2) for k, v in d['split_ms_per_iter'].items()})"; done | tee $out/contact_ab.txt | cut -c1-300
