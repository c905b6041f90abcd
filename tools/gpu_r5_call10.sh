#!/bin/bash
# Round 5, tenth GPU call: test_factorize_solve[1] (the rocSOLVER comparison back end) failed once when tests/test_gpu_sharded.py ran in front of it in the same
# process -- alone, in the suite's own order, and in that order again.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r5c10
mkdir -p $out
f() { grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | grep -E "passed|failed|Error|error|FAILED" | tail -5; }
echo "--- alone"; timeout 300 python -m pytest "tests/test_gpu_parity.py::test_factorize_solve" -m gpu -q -x 2>&1 | f
echo "--- parity file"; timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | f
echo "--- rccl binding test first"; timeout 300 python -m pytest "tests/test_gpu_sharded.py::test_rccl_binding_from_c_on_one_rank" "tests/test_gpu_parity.py::test_factorize_solve" -m gpu -q -x 2>&1 | f
echo "--- sharded file first"; timeout 600 python -m pytest tests/test_gpu_sharded.py "tests/test_gpu_parity.py::test_factorize_solve" -m gpu -q -x 2>&1 | f
