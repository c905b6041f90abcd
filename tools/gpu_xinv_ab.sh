#!/bin/bash
# A/B: the explicit inverses of the top levels (the tail of the factorisation) against block-by-block triangle sweeps there.  Outputs: gpurun_out/$1/
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r03x2}
mkdir -p $out
for rep in 1 2; do
for k in 0 1 2 3; do
    IPCGPU_MF_XINV_SKIP_TOP=$k timeout 300 python bench.py --no-cpu-baseline --no-contact > $out/bench_skip${k}_$rep.json 2>> $out/err.log
done
done
python - <<PY
import json
for k in (0, 1, 2, 3):
    r = []
    for rep in (1, 2):
        try:
            d = json.load(open("$out/bench_skip%d_%d.json" % (k, rep))); r.append("%.1f it/s (factor %.3f solve %.3f)" % (d["value"], d["solver"]["factor_ms"], d["solver"]["solve_ms"]))
        except Exception as e:
            r.append("failed %s" % e)
    print("skip top", k, " | ".join(r))
PY
( IPCGPU_MF_XINV_SKIP_TOP=1 timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4 ) | tee $out/tests_skip1.txt
