#!/bin/bash
# The GPU tests added last + an SQ-counter pass over the solver kernels (tools/pmc_solver.py).  Outputs: gpurun_out/$1/
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-r03u}
mkdir -p $out
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl'
( timeout 600 python -m pytest tests/test_gpu_vs_reference.py tests/test_main_hip.py -m gpu -q -k "abs_parameters or global_time_range or absolute_parameters" 2>&1 | grep -v "$F" | tail -15 ) | tee $out/new_gpu_tests.txt
cd /tmp
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_WAVES \
    --kernel-trace --output-format csv -d $out/pmc_sq -- python $R/tools/pmc_solver.py workload > $out/pmc_sq.log 2>&1
if ! ls $out/pmc_sq/*/*counter_collection.csv > /dev/null 2>&1; then
    echo "first counter set refused; retrying with three" | tee -a $out/pmc_sq.log
    tail -5 $out/pmc_sq.log
    rm -rf $out/pmc_sq
    timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $out/pmc_sq -- python $R/tools/pmc_solver.py workload >> $out/pmc_sq.log 2>&1
fi
cd $R
python tools/pmc_solver.py parse $out/pmc_sq > $out/pmc_solver_sq.json 2>> $out/pmc_sq.log
grep -v "$F" $out/pmc_sq.log | tail -4
python - <<PY
import json
d = json.load(open("$out/pmc_solver_sq.json"))
for k, r in d["kernels"].items():
    print(k, r["dispatches"], {a: round(b, 3) for a, b in r.items() if isinstance(b, float)})
PY
# keep the merged output small: the raw csv files are large
find $out/pmc_sq -name "*.csv" -size +8M -delete
