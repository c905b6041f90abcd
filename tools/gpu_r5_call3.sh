#!/bin/bash
# Round 5, third GPU call: (1) the whole GPU suite (the sharded solver now sends point to point; fused-eligible fronts of mixed levels ride on the batched
# launches), (2) A/B of that routing on the headline, the contact bench and mat433, (3) rocprofv3 kernel tables at mat433 and of the contact bench,
# (4) two and four ranks on this one device over gloo (plumbing: bytes on the wire).
#   gpurun --timeout 1500 -- 'bash tools/gpu_r5_call3.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=gpurun_out/r5c3
mkdir -p $out
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -12 ) | tee $out/gpu_tests.txt
echo "=== mat150 (+ contact sub-records)"
bash tools/gpu_ab.sh r5c3/ab150 "" "-" "IPCGPU_MF_MIXED_LEVELS=1"
echo "=== mat433"
STEPS=12 bash tools/gpu_ab.sh r5c3/ab433 "--no-contact --size 433" "-" "IPCGPU_MF_MIXED_LEVELS=1"
echo "=== contact bench"
for rep in 1 2; do for s in "X=0" "IPCGPU_MF_MIXED_LEVELS=1"; do env $s timeout 300 python tools/bench_contact.py --n 100 --steps 12 2>/dev/null | python -c "
import sys, json
d = json.load(sys.stdin); print('$s', round(d['ms_per_iter_wall'], 3), d['newton_iterations'], {k: round(v, 2) for k, v in d['split_ms_per_iter'].items()})"; done; done | tee $out/contact_ab.txt
echo "=== kernel tables"
rm -rf /tmp/prof433 /tmp/profc
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof433 -o run -- python $R/bench.py --no-cpu-baseline --no-large --no-contact --size 433 --steps 12 --warmup 3 > $R/$out/bench433_under_rocprof.json 2> /dev/null )
db=$(find /tmp/prof433 -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db $out/mat433_kernel_stats.md > /dev/null && head -22 $out/mat433_kernel_stats.md | cut -c1-120
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/profc -o run -- python $R/tools/bench_contact.py --n 100 --steps 12 > $R/$out/contact_under_rocprof.json 2> /dev/null )
db=$(find /tmp/profc -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db $out/contact_kernel_stats.md > /dev/null && head -22 $out/contact_kernel_stats.md | cut -c1-120
echo "=== ranks on one device (gloo plumbing)"
for n in 2 4; do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2953$n bench.py --gpus $n --steps 20 --warmup 3 --single-device-test --large-size 0 2> $out/sd$n.err | tail -1 > $out/single_device_$n.json
  python - <<PY
import json
try:
    d = json.load(open("$out/single_device_$n.json"))
    print($n, "ranks:", round(d["value"], 1), "it/s;", {k: (round(v) if isinstance(v, (int, float)) else v) for k, v in d["comm_per_iter"].items()})
except Exception as e:
    print($n, "ranks: ERR", e); print(open("$out/sd$n.err").read()[-1500:])
PY
done
