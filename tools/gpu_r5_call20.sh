#!/bin/bash
# Round 5, twentieth GPU call: bench.py launched as the driver launches it for N = 2 and N = 4, all ranks on this ONE device (--single-device-test: gloo through a host
# bounce, so the rates mean nothing) -- the JSON lines with their communication records and the >= 1 M-tet second workload, on the final code.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r5c20
mkdir -p $out
for n in 2 4; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2955$n bench.py --gpus $n --steps 20 --warmup 3 --single-device-test 2> $out/sd$n.err | tail -1 > $out/single_device_$n.json
  python - <<PY
import json
try:
    d = json.load(open("$out/single_device_$n.json"))
    print("N=$n", round(d["value"], 1), "it/s |", d["transport"], "|", {k: (round(v) if isinstance(v, float) else v) for k, v in d["comm_per_iter"].items()})
    lw = d.get("large_workload") or {}
    print("   large_workload:", lw.get("workload"), round(lw.get("value", 0), 2), "it/s, shared flops", round(lw.get("shared_flop_fraction", 0), 3), lw.get("comm_per_iter_rank0"))
except Exception as e:
    print("N=$n failed", e); print(open("$out/sd$n.err").read()[-1500:])
PY
done
