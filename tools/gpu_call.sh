#!/bin/bash
# One parametrised GPU call (replaces the per-call scripts of round 5).  Runs, in order, whatever the environment asks for and writes under gpurun_out/<name>/:
#   TESTS="tests/test_gpu_contact.py ..."   pytest -m gpu on those files (TESTS=all: the whole suite); K="expr" adds -k
#   CONTACT=1        tools/bench_contact.py (2 x mat100) plain + under rocprofv3 --kernel-trace --stats -> contact.json, contact_kernel_stats.md
#   LARGE="250 3"    tools/bench_contact.py at that size (contact_large) + kernel stats                -> contact_large.json, contact_large_kernel_stats.md
#   TWIST=1          tools/bench_mat_twist.py                                                          -> mat_twist.json
#   BENCH="flags"    bench.py with those flags                                                         -> bench.json  (BENCH_PROF=1: also under rocprofv3)
#   CMD="..."        anything else, last
# usage: gpurun --timeout 1500 -- 'NAME=r6_call1 TESTS=all CONTACT=1 bash tools/gpu_call.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/${NAME:-call}
mkdir -p $out
filt() { grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl"; }
prof() { # prof <tag> <cmd...>: kernel table of a command
  tag=$1; shift
  rm -rf /tmp/prof_$tag
  ( cd /tmp && timeout ${PROF_TIMEOUT:-600} rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o $tag -- "$@" > $GRAFT_REPO_ROOT/$out/${tag}_under_rocprof.json 2>> $GRAFT_REPO_ROOT/$out/err.log )
  db=$(find /tmp/prof_$tag -name "*.db" | head -1)
  if [ -n "$db" ]; then
    python tools/rocprof_summary.py $db $out/${tag}_kernel_stats.md > /dev/null
    case $tag in
      contact*) python tools/contact_timeline.py $db ${TIMELINE_ITER:-20} > $out/${tag}_timeline.txt 2>&1;;
      bench) python tools/rocprof_timeline.py $db 30 > $out/${tag}_timeline.txt 2>&1;;
    esac
  fi
}
if [ -n "$TESTS" ]; then
  t="$TESTS"; [ "$TESTS" = "all" ] && t="tests"
  ( timeout ${TEST_TIMEOUT:-1200} python -m pytest $t -m gpu -q -x --durations=12 ${K:+-k "$K"} 2>&1 | filt | tail -${TEST_TAIL:-30} ) | tee $out/tests.txt
fi
if [ -n "$CONTACT" ]; then
  timeout 300 python tools/bench_contact.py --n 100 --layers 2 --steps 12 --max-iter 12 > $out/contact.json 2>> $out/err.log
  python - $out/contact.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("contact: %.3f ms/iter, %d iterations" % (d["ms_per_iter_wall"], d["newton_iterations"]), {k[:24]: round(v, 3) for k, v in d["split_ms_per_iter"].items()})
PY
  [ -n "$NOPROF" ] || prof contact python $GRAFT_REPO_ROOT/tools/bench_contact.py --n 100 --layers 2 --steps 12 --max-iter 12
fi
if [ -n "$LARGE" ]; then
  set -- $LARGE
  timeout 900 python tools/bench_contact.py --n $1 --layers $2 --steps ${LARGE_STEPS:-3} --max-iter ${LARGE_ITERS:-8} > $out/contact_large.json 2>> $out/err.log
  python - $out/contact_large.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("contact_large: %.3f ms/iter, %d iterations, %d nodes %d tets" % (d["ms_per_iter_wall"], d["newton_iterations"], d["n_nodes"], d["n_tets"]), {k[:24]: round(v, 3) for k, v in d["split_ms_per_iter"].items()},
      [(c["nActive"], c["nPara"], c["nPatternChanges"]) for c in d["contact_state_per_step"]], "precompute %.1f s" % d["precompute_s"])
PY
  [ -n "$NOPROF" ] || prof contact_large python $GRAFT_REPO_ROOT/tools/bench_contact.py --n $1 --layers $2 --steps ${LARGE_STEPS:-3} --max-iter ${LARGE_ITERS:-8}
fi
if [ -n "$TWIST" ]; then
  timeout 900 python tools/bench_mat_twist.py ${TWIST_FLAGS} > $out/mat_twist.json 2>> $out/err.log
  python - $out/mat_twist.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k in ("early", "wrapped"):
    r = d[k]
    if r.get("value") is None:
        print(k, r)
        continue
    print("mat_twist %s: steps %s, %.3f ms/iter, its %s, active %d para %d cand %d" % (k, r["window_steps"], r["ms_per_iter"], r["iterations_per_step"], r["active_constraints_at_end"], r["mollified_at_end"],
          r["candidates_at_end"]), {kk[:24]: round(v, 3) for kk, v in r["split_ms_per_iter"].items()})
PY
fi
if [ -n "$BENCH" ]; then
  [ "$BENCH" = "default" ] && BENCH=""
  timeout 900 python bench.py $BENCH > $out/bench.json 2>> $out/err.log
  python - $out/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
c = d.get("contact") or {}
print("bench: %.1f it/s  factor %.3f solve %.3f asm %.4f ms  contact %s ms/iter" % (d["value"], d["solver"]["factor_ms"], d["solver"]["solve_ms"], d["roofline"]["avg_launch_ms"], c.get("ms_per_iter")))
for k in ("rods_twist", "sphere_on_mat", "roofline_large", "contact_large", "mat_twist_as_shipped"):
    if k in d:
        print("  ", k, json.dumps(d[k])[:400])
PY
  if [ -n "$BENCH_PROF" ]; then prof bench python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-contact --no-large $BENCH; fi
fi
if [ -n "$CMD" ]; then bash -c "$CMD" 2>&1 | filt | tail -${CMD_TAIL:-40}; fi
tail -5 $out/err.log 2>/dev/null || true
