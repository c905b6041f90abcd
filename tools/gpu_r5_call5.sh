#!/bin/bash
# Round 5, fifth GPU call: (1) the solver-level GPU tests, then the whole suite, on: the entries of A added by the extend-add kernel (no k_scatter_big launch),
# the Schur complement of the single-workgroup fronts on the matrix cores, the atexit summary of the percall adapter; (2) A/B of the two solver changes
# against the library of the commit before (ipc_amd/libipcgpu_prev.so); (3) the leaf size of the dissection swept again (the narrow-panel exit of the pivot
# sweep and the routing of mixed levels moved the balance since round 4).
#   gpurun --timeout 1500 -- 'bash tools/gpu_r5_call5.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r5c5
mkdir -p $out
timeout 120 python tools/check_solver.py 2>&1 | tail -1
( timeout 1000 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -40 ) | tee $out/gpu_tests.txt | tail -12
echo "=== mat150 (+ contact sub-records)"
bash tools/gpu_ab.sh r5c5/ab150 "" "-" "IPCGPU_LIB_VARIANT=prev"
echo "=== mat433"
STEPS=12 bash tools/gpu_ab.sh r5c5/ab433 "--no-contact --size 433" "-" "IPCGPU_LIB_VARIANT=prev"
echo "=== contact bench"
for rep in 1 2; do for s in "X=0" "IPCGPU_LIB_VARIANT=prev"; do env $s timeout 300 python tools/bench_contact.py --n 100 --steps 12 2>/dev/null | python -c "
import sys, json
d = json.load(sys.stdin); print('$s', round(d['ms_per_iter_wall'], 3), d['newton_iterations'], {k: round(v, 2) for k, v in d['split_ms_per_iter'].items()})"; done; done | tee $out/contact_ab.txt | cut -c1-250
echo "=== leaf size"
run() {
  env $1 timeout 60 python bench.py --no-cpu-baseline --no-large --no-contact --steps 150 --warmup 10 2>/dev/null | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('%-24s %7.1f it/s  factor %.3f  sweeps %.3f' % ('$1', d['value'], d['solver']['factor_ms'], d['solver']['solve_ms']))
except Exception as e:
    print('%-24s failed: %s' % ('$1', e))
" | tee -a $out/leaf.txt
}
for v in 12 9 10 11 13 14 12; do run "IPCGPU_ND_LEAF=$v"; done
