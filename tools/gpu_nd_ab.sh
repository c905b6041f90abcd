#!/bin/bash
# host-side analysis of a contact pattern change: parallel depth / minimum size of the dissection's threaded levels (library variants of -DMF_ND_PAR_DEPTH / -DMF_ND_PAR_MIN)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
nproc
for rep in 1 2; do
for v in "" pd4 pd5; do
  IPCGPU_LIB_VARIANT=$v IPCGPU_MF_SETUP_TIMES=1 timeout 300 python tools/bench_contact.py --n 100 --layers 2 --steps 12 --max-iter 12 2> /tmp/t_$v.txt > /tmp/c_$v.json
  python - "$v" /tmp/t_$v.txt /tmp/c_$v.json <<'PY'
import sys, json, re, collections
v, tf, cf = sys.argv[1:]
acc = collections.defaultdict(list)
for l in open(tf):
    m = re.match(r"mf (analyze|setup) (.*?)\s+([\d.]+) ms", l)
    if m: acc[m.group(2).strip()].append(float(m.group(3)))
d = json.load(open(cf))
# the steady state: drop the first analysis (allocations)
print("[%s] %.3f ms/iter symbolic %.3f | " % (v, d["ms_per_iter_wall"], d["split_ms_per_iter"]["pattern_change:symbolic_analysis"]) + "  ".join("%s %.2f" % (k[:18], sum(x[1:]) / max(len(x) - 1, 1)) for k, x in acc.items()))
PY
done
done
