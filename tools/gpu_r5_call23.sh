#!/bin/bash
# Round 5, twenty-third GPU call: the new test that forces the two-level blocking on at test size.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "two_level or factorize_solve" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8
