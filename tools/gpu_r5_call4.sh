#!/bin/bash
# Round 5, fourth GPU call: (1) the whole GPU suite, every failure listed (new: role A of the step kernel on the matrix cores, self-contact on the device in
# HipOptimizer's percall mode, chain10 on the envelope criterion); (2) A/B of role A: the product library against a variant built with the scalar
# 4 x 4 register tiles of rounds 1-4 (ipc_amd/libipcgpu_fmarolea.so), at 45 K and 375 K nodes and on the contact bench.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r5_call4.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r5c4
mkdir -p $out
( timeout 1000 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -40 ) | tee $out/gpu_tests.txt | tail -25
echo "=== mat150 (+ contact sub-records)"
bash tools/gpu_ab.sh r5c4/ab150 "" "-" "IPCGPU_LIB_VARIANT=fmarolea"
echo "=== mat433"
STEPS=12 bash tools/gpu_ab.sh r5c4/ab433 "--no-contact --size 433" "-" "IPCGPU_LIB_VARIANT=fmarolea"
echo "=== contact bench"
for rep in 1 2; do for s in "X=0" "IPCGPU_LIB_VARIANT=fmarolea"; do env $s timeout 300 python tools/bench_contact.py --n 100 --steps 12 2>/dev/null | python -c "
import sys, json
d = json.load(sys.stdin); print('$s', round(d['ms_per_iter_wall'], 3), d['newton_iterations'], {k: round(v, 2) for k, v in d['split_ms_per_iter'].items()})"; done; done | tee $out/contact_ab.txt
