#!/bin/bash
# HBM traffic of the assembly kernel from the PMC counters (FETCH_SIZE and WRITE_SIZE in separate passes, each calibrated on a 1 GiB copy: tools/pmc_traffic.py) at mat150 and mat433
#   usage: gpurun -- 'NAME=r6_pmc bash tools/gpu_pmc_traffic.sh'   -> gpurun_out/$NAME/pmc_assembly_traffic_<size>.json
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/${NAME:-pmc}
mkdir -p $out
for size in ${PMC_SIZES-150 433}; do
  rm -rf $out/pmc_rd_$size $out/pmc_wr_$size
  ( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$out/pmc_rd_$size -- python $R/tools/pmc_traffic.py workload $size > /dev/null 2>&1 )
  ( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$out/pmc_wr_$size -- python $R/tools/pmc_traffic.py workload $size > /dev/null 2>&1 )
  python tools/pmc_traffic.py parse $out/pmc_rd_$size $out/pmc_wr_$size $size > $out/pmc_assembly_traffic_$size.json 2> $out/pmc_$size.err
  rm -rf $out/pmc_rd_$size $out/pmc_wr_$size
  python - $out/pmc_assembly_traffic_$size.json $size <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("pmc", sys.argv[2], round(d["traffic_bytes"] / 1e6, 1), "MB vs", round(d["algorithmic_bytes"] / 1e6, 1), "MB algorithmic:", round(d["traffic_over_algorithmic"], 3))
PY
done
