#!/bin/bash
# A/B inside one call: (1) where a pattern change spends its host time (IPCGPU_PATTERN_TIMES), (2) the explicit inverse of a 1 440-column root by bordering inside the
# step launches (library variant b1536: -DMF_BORDER_MAX_NC=1536) against recursive doubling behind them (the default, 1024)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/${NAME:-border_ab}
mkdir -p $out
IPCGPU_PATTERN_TIMES=1 timeout 300 python tools/bench_contact.py --n 100 --layers 2 --steps 12 --max-iter 12 > $out/contact_times.json 2> $out/pattern_times.txt
grep "pattern change" $out/pattern_times.txt | awk '{k=$3" "$4" "$5" "$6; gsub(/[0-9.]+ ms/,"",k); t[$0]=1} {print}' | tail -40
for rep in 1 2; do
for v in "" b1536; do
  IPCGPU_LIB_VARIANT=$v timeout 300 python tools/bench_contact.py --n 100 --layers 2 --steps 12 --max-iter 12 > $out/contact_${v:-default}_$rep.json 2>> $out/err.log
  python - $out/contact_${v:-default}_$rep.json "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("variant [%s]: %.3f ms/iter, factor %.3f ms solve %.3f ms, split factor+sweeps %.3f" % (sys.argv[2], d["ms_per_iter_wall"], d["solver"]["factor_ms"], d["solver"]["solve_ms"], d["split_ms_per_iter"]["factorisation+triangular_sweeps"]))
PY
done
done
