#!/bin/bash
# Contact-path A/B on the GPU box: parity tests of the contact rows, the contact benchmark with the round-3 kernels and with the
# round-2 variants they replace (IPCGPU_HESS_LDS, IPCGPU_CCD_TWO_PASS), and the kernel table of the benchmark.  Outputs: gpurun_out/$1/
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r03i}
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then
  ( timeout 1200 python -m pytest tests/test_gpu_contact.py tests/test_gpu_vs_reference.py tests/test_gpu_friction.py tests/test_gpu_sharded.py tests/test_gpu_scenes.py -m gpu -q -x 2>&1 | tail -8 ) > $out/gpu_tests.txt; cat $out/gpu_tests.txt
fi
timeout 300 python tools/bench_contact.py --n 100 --steps 12 > $out/contact_new.json 2> $out/contact_new.err
IPCGPU_HESS_LDS=1 IPCGPU_CCD_TWO_PASS=1 timeout 300 python tools/bench_contact.py --n 100 --steps 12 > $out/contact_old_variants.json 2> /dev/null
rm -rf /tmp/profc
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/profc -o run -- python $R/tools/bench_contact.py --n 100 --steps 12 > $R/$out/contact_under_rocprof.json 2> /dev/null )
db=$(find /tmp/profc -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py $db $out/contact_kernel_stats.md > /dev/null
python - <<PY
import json
for f in ("contact_new", "contact_old_variants"):
    try:
        d = json.load(open("$out/%s.json" % f))
        print(f, round(d["ms_per_iter_wall"], 3), d["newton_iterations"], {k: round(v, 2) for k, v in d["split_ms_per_iter"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
head -24 $out/contact_kernel_stats.md | cut -c1-110
for pr in 0 1 2; do IPCGPU_HESS_PROBE=$pr timeout 120 python tools/gpu_hess_probe.py 2>&1 | tail -1; done | tee $out/hess_probe.txt
