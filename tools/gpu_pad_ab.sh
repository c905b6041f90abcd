#!/bin/bash
# A/B of the contact pattern's look-ahead (ipcgpu_opt_set_pattern_lookahead) on the three contact workloads of the bench line, one gpurun call.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/${NAME:-pad_ab}
mkdir -p $out
for pad in ${PADS:-0.5 1 2.25 4}; do
  timeout 300 python tools/bench_contact.py --n 100 --layers 2 --steps 12 --max-iter 12 --pad $pad > $out/contact_pad$pad.json 2>> $out/err.log
  timeout 300 python tools/bench_mat_twist.py --pad $pad > $out/twist_pad$pad.json 2>> $out/err.log
  [ -n "$NOLARGE" ] || timeout 600 python tools/bench_contact.py --n 250 --layers 3 --steps 2 --max-iter 4 --pad $pad > $out/large_pad$pad.json 2>> $out/err.log
done
python - $out ${PADS:-0.5 1 2.25 4} <<'PY'
import json, sys
out, pads = sys.argv[1], sys.argv[2:]
for pad in pads:
    line = "pad %-5s" % pad
    for name in ("contact", "large"):
        try:
            d = json.load(open(f"{out}/{name}_pad{pad}.json")); s = d["split_ms_per_iter"]
            line += " | %s %.2f ms/it (factor %.2f, analysis %.2f+%.2f, %d changes, %.0f GF)" % (name, d["ms_per_iter_wall"], s["factorisation+triangular_sweeps"], s["pattern_change:symbolic_analysis"],
                      s["pattern_change:set_pattern"], d["contact_state_per_step"][-1]["nPatternChanges"], d["solver"]["factor_gflop"])
        except Exception as e:
            line += " | %s -" % name
    try:
        d = json.load(open(f"{out}/twist_pad{pad}.json")); w = d["wrapped"]; s = w["split_ms_per_iter"]
        line += " | twist wrapped %.2f ms/it (factor %.2f, analysis %.2f+%.2f, %d changes, its %s) early %.2f" % (w["ms_per_iter"], s["factorisation+triangular_sweeps"], s["pattern_change:symbolic_analysis"],
                  s["pattern_change:set_pattern"], w["pattern_changes_so_far"], w["iterations_per_step"], d["early"]["ms_per_iter"])
    except Exception as e:
        line += " | twist -"
    print(line)
PY
tail -3 $out/err.log 2>/dev/null
