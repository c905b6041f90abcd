#!/bin/bash
# End-of-round check on the GPU box: smoke(), the whole GPU suite, the bench line.  Outputs: gpurun_out/$1/
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r03z}
mkdir -p $out
export TMPDIR=/tmp
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3 ) | tee $out/smoke.txt
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -12 ) | tee $out/gpu_tests.txt
timeout 600 python bench.py > $out/bench_line.json 2> $out/bench.err
python - <<PY
import json
d = json.load(open("$out/bench_line.json"))
print("bench", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 4), "ms; roofline frac", round(d["roofline"]["frac"], 4), "cpu", d["cpu_baseline"]["value"], "contact", d["contact"]["ms_per_iter"])
PY
