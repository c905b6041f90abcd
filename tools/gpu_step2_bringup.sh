#!/bin/bash
# First GPU call of the next round: bring-up of k_big_step2 (two panels per step launch, IPCGPU_MF_STEP2=1; written and emulated on the host at the end of
# round 4, never run).  Stages, each only if the one before passed:
#   1. residuals + not-PD detection of the solver alone (tools/check_solver.py), first without the bordering role (IPCGPU_MF_XINV_BORDER=0: roles A', B', M only),
#      then with it (role C');
#   2. the solver-level GPU tests under the switch;
#   3. the headline A/B (two interleaved runs each).
#   gpurun --timeout 900 -- 'bash tools/gpu_step2_bringup.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/step2
mkdir -p $out
export TMPDIR=/tmp
ok() { python - "$1" <<'PY'
import re, sys
m = re.search(r"factorize=True/True residuals ([0-9.e+-]+) ([0-9.e+-]+) notPD_detected=(\w+)", open(sys.argv[1]).read())
sys.exit(0 if m and float(m.group(1)) < 1e-10 and float(m.group(2)) < 1e-10 and m.group(3) == "True" else 1)
PY
}
IPCGPU_MF_STEP2=1 IPCGPU_MF_XINV_BORDER=0 timeout 120 python tools/check_solver.py > $out/check_noborder.txt 2>&1; tail -1 $out/check_noborder.txt
ok $out/check_noborder.txt || { echo "STAGE 1a FAILED (roles A', B', M)"; exit 0; }
IPCGPU_MF_STEP2=1 timeout 120 python tools/check_solver.py > $out/check_border.txt 2>&1; tail -1 $out/check_border.txt
ok $out/check_border.txt || { echo "STAGE 1b FAILED (role C')"; exit 0; }
( IPCGPU_TEST_STEP2=1 timeout 300 python -m pytest tests/test_gpu_step2.py -m gpu -q -x; IPCGPU_MF_STEP2=1 timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q -x ) 2>&1 | grep -E "passed|failed|Error" | tail -3 | tee $out/tests.txt
grep -q failed $out/tests.txt && { echo "STAGE 2 FAILED"; exit 0; }
CHECK=1 bash tools/gpu_ab.sh step2/ab "" "IPCGPU_MF_STEP2=0" "IPCGPU_MF_STEP2=1"
STEPS=12 bash tools/gpu_ab.sh step2/ab433 "--no-contact --size 433" "IPCGPU_MF_STEP2=0" "IPCGPU_MF_STEP2=1"
