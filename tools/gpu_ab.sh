#!/bin/bash
# Generic A/B of environment settings on the headline bench inside ONE gpurun call (same box): every setting twice, interleaved.
#   tools/gpu_ab.sh <outdir> "<bench flags>" "VAR=a" "VAR=b OTHER=c" ...      ("-" = no setting)
# Prints it/s, factor / solve ms per setting.  Optional: CHECK=1 runs tools/check_solver.py (residuals, not-PD detection) under every setting; TESTS="tests/test_gpu_fullsize.py ..." runs those first.
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1
flags=$2
shift 2
mkdir -p $out
export TMPDIR=/tmp
if [ -n "$TESTS" ]; then
  ( timeout ${TEST_TIMEOUT:-900} python -m pytest $TESTS -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6 ) | tee $out/tests.txt
fi
i=0
for rep in 1 2; do
  i=0
  for setting in "$@"; do
    i=$((i + 1))
    if [ "$setting" = "-" ]; then env_s=""; else env_s="$setting"; fi
    if [ -n "$CHECK" ] && [ $rep = 1 ]; then ( env $env_s timeout 120 python tools/check_solver.py 2>&1 | tail -1 ) | sed "s/^/[$setting] /"; fi
    env $env_s timeout 300 python bench.py --no-cpu-baseline --no-large --steps ${STEPS:-200} --warmup 10 $flags > $out/ab_${i}_$rep.json 2>> $out/err.log
  done
done
python - "$out" "$@" <<'PY'
import json, sys
out, settings = sys.argv[1], sys.argv[2:]
for i, s in enumerate(settings, 1):
    r = []
    for rep in (1, 2):
        try:
            d = json.load(open(f"{out}/ab_{i}_{rep}.json"))
            c = d.get("contact") or {}
            r.append("%.1f it/s (factor %.3f solve %.3f asm %.4f%s)" % (d["value"], d["solver"]["factor_ms"], d["solver"]["solve_ms"], d["roofline"]["avg_launch_ms"],
                                                                  (" contact %.2f ms/it" % c["ms_per_iter"]) if c.get("ms_per_iter") else ""))
        except Exception as e:
            r.append("failed %s" % e)
    print("%-40s %s" % (s, " | ".join(r)))
PY
tail -5 $out/err.log 2>/dev/null
