#!/bin/bash
# Round 5, fifteenth GPU call: at 375 K nodes the step launches of the levels whose fronts border their explicit inverse (k_big_step<true>) average 52 us -- is it the
# bordering role?  The widest bordered front swept at that size (it was only ever swept at 45 K nodes and on the contact stack), with and without a lower threshold for
# the two-level blocking.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
echo "=== mat433"
STEPS=12 bash tools/gpu_ab.sh r5c15/ab433 "--no-contact --size 433" "-" "IPCGPU_MF_BORDER_MAX_NC=768" "IPCGPU_MF_BORDER_MAX_NC=512" "IPCGPU_MF_BORDER_MAX_NC=384" "IPCGPU_MF_BORDER_MAX_NC=0" "IPCGPU_MF_BORDER_MAX_NC=512 IPCGPU_MF_BULK_MIN_NC=512" "IPCGPU_MF_BULK_MIN_NC=768"
