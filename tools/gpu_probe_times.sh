#!/bin/bash
# Where a pattern change spends its time on the contact benchmark (host-side laps, stderr), the contact benchmark itself, the headline bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r03n}
mkdir -p $out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_contact.py tests/test_gpu_vs_reference.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -4 ) > $out/gpu_tests.txt; cat $out/gpu_tests.txt
timeout 300 python tools/bench_contact.py --n 100 --steps 12 > $out/contact.json 2> /dev/null
IPCGPU_PATTERN_TIMES=1 IPCGPU_MF_SETUP_TIMES=1 timeout 300 python tools/bench_contact.py --n 100 --steps 12 > /dev/null 2> $out/pattern_times.txt
timeout 300 python bench.py --no-cpu-baseline --no-contact > $out/bench_line.json 2> /dev/null
python - <<PY
import json
d = json.load(open("$out/contact.json"))
print("contact", round(d["ms_per_iter_wall"], 3), d["newton_iterations"], {k: round(v, 2) for k, v in d["split_ms_per_iter"].items()})
d = json.load(open("$out/bench_line.json"))
print("bench", round(d["value"], 1), d["ms_per_step"], d["config"].get("split_ms_per_iter"))
PY
grep -v "^\[" $out/pattern_times.txt | head -150
