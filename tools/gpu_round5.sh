#!/bin/bash
# Round-5 evidence on the GPU box (outputs under gpurun_out/r05/, copied into profiles/ afterwards):
#   full GPU test suite, bench line (CPU baseline + contact sub-record), rocprofv3 kernel table + timeline of the same command,
#   the 1.12 M-tet size, the contact benchmark's kernel table, and the PMC traffic of the assembly kernel at mat150 and mat433
#   (FETCH_SIZE and WRITE_SIZE in separate passes, each calibrated on a 1 GiB copy: tools/pmc_traffic.py).
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then
  ( timeout 1500 python -m pytest tests -m gpu -q --tb=line -rf 2>&1 | grep -E "FAILED|passed|failed|assert" | cut -c1-400 | tail -12 ) > $out/gpu_tests.txt; cat $out/gpu_tests.txt
fi
timeout 400 python bench.py > $out/bench_line.json 2> $out/bench.err; grep bench $out/bench.err | tail -12
rm -rf /tmp/prof_r05
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_r05 -o run -- python $R/bench.py --no-cpu-baseline --no-contact --no-large --steps 60 --warmup 10 > $R/$out/bench_line_under_rocprof.json 2> /dev/null )
db=$(find /tmp/prof_r05 -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py $db $out/kernel_stats.md > /dev/null && python tools/rocprof_timeline.py $db 12 > $out/timeline.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-contact --no-large --size 433 --steps 12 --warmup 3 > $out/bench_mat433.json 2> /dev/null
rm -rf /tmp/profc_r05
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/profc_r05 -o run -- python $R/tools/bench_contact.py --n 100 --steps 12 > $R/$out/contact_bench_under_rocprof.json 2> /dev/null )
db=$(find /tmp/profc_r05 -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py $db $out/contact_kernel_stats.md > /dev/null
for size in ${PMC_SIZES-150 433}; do
  rm -rf $out/pmc_rd_$size $out/pmc_wr_$size
  ( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$out/pmc_rd_$size -- python $R/tools/pmc_traffic.py workload $size > /dev/null 2>&1 )
  ( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$out/pmc_wr_$size -- python $R/tools/pmc_traffic.py workload $size > /dev/null 2>&1 )
  python tools/pmc_traffic.py parse $out/pmc_rd_$size $out/pmc_wr_$size $size > $out/pmc_assembly_traffic_$size.json 2> $out/pmc_$size.err
  rm -rf $out/pmc_rd_$size $out/pmc_wr_$size
done
python - <<PY
import json
for f in ("bench_line", "bench_mat433"):
    try:
        d = json.load(open("$out/%s.json" % f))
        print(f, round(d["value"], 1), {k: round(v, 3) for k, v in d["split_ms_per_iter"].items()}, "asm", round(d["roofline"]["avg_launch_ms"], 4), round(d["roofline"]["frac"], 4),
              [round(r["frac"], 4) for r in d.get("roofline_solver", [])], d.get("cpu_baseline", {}).get("value"), (d.get("contact") or {}).get("ms_per_iter"))
    except Exception as e:
        print(f, "ERR", e)
for s in (150, 433):
    try:
        d = json.load(open("$out/pmc_assembly_traffic_%d.json" % s)); print("pmc", s, round(d["traffic_bytes"] / 1e6, 1), "MB vs", round(d["algorithmic_bytes"] / 1e6, 1), "MB algorithmic:", round(d["traffic_over_algorithmic"], 3))
    except Exception as e:
        print("pmc", s, "ERR", e)
PY
head -12 $out/kernel_stats.md; head -12 $out/contact_kernel_stats.md
