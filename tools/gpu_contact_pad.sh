#!/bin/bash
# A/B of environment settings on the contact benchmark (2 x mat100 stack) inside one gpurun call: tools/gpu_contact_pad.sh <outdir> "VAR=a" ...
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; shift
mkdir -p $out
i=0
for setting in "$@"; do
  i=$((i + 1))
  if [ "$setting" = "-" ]; then env_s=""; else env_s="$setting"; fi
  for rep in 1 2; do
    env $env_s timeout 200 python tools/bench_contact.py --n 100 --steps 12 > $out/c_${i}_$rep.json 2>> $out/err.log
  done
done
python - "$out" "$@" <<'PY'
import json, sys
out, settings = sys.argv[1], sys.argv[2:]
for i, s in enumerate(settings, 1):
    r = []
    for rep in (1, 2):
        try:
            d = json.load(open(f"{out}/c_{i}_{rep}.json"))
            sp = d["split_ms_per_iter"]
            r.append("%.2f ms/it (%d its, %d changes; factor %.2f, analysis %.2f, sets %.2f, asm %.2f)" % (d["ms_per_iter_wall"], d["newton_iterations"], d["contact_state_per_step"][-1]["nPatternChanges"],
                     sp["factorisation+triangular_sweeps"], sp["pattern_change:symbolic_analysis"] + sp["pattern_change:set_pattern"], sp["constraint_sets"],
                     sp["assembly+barrier_hessian(+host connectivity of a pattern change)"]))
        except Exception as e:
            r.append("failed %s" % e)
    print("%-36s %s" % (s, " | ".join(r)))
PY
