"""SQ counters of the solver kernels (MI355X_MICROARCH.md, rocprofv3 PMC section): where the waves of the factorisation spend their cycles.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_WAVES \
      --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -- python $R/tools/pmc_solver.py workload
  python tools/pmc_solver.py parse gpurun_out/pmc_sq > profiles/r03_pmc_solver_sq.json

Units (guide): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES cycles summed over
SIMDs, SQ_BUSY_CU_CYCLES quad-cycles summed over CUs.  The ratios below stay inside one unit:
  wait_any / wave_cycles          waves parked on s_waitcnt or a barrier (memory latency, the pivot chain's __syncthreads)
  wait_inst_any / wave_cycles     issue stalls (MFMA read-after-write, busy pipe)
  active_inst_any / wave_cycles   issuing
Every dispatch is attributed to its kernel (the solver replays no graphs)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ["k_big_step", "k_big_schur64", "k_big_schur", "k_front_fused", "k_extend_add", "k_xinv_gemm", "k_fwd_level", "k_bwd_level", "k_big_fwd_rect",
           "k_big_bwd_init", "k_assemble_patch"]


def workload(size=150):
    sys.path.insert(0, ROOT)
    from ipc_amd import lib, scene
    V, F = scene.make_mat(size)
    c = lib.Context(0)
    c.set_mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
    c.opt_init(0.04, False)
    left, right = scene.border_verts(V, 0.01)
    c.set_twist(left, right)
    c.set_positions(scene.twist_state(scene.jitter(V, F), 0.3))
    c.set_pattern()
    c.bench_assembly(0.04 ** 2, 2)
    c.analyze_pattern()
    print("factor / solve ms (counter collection serialises the kernels):", c.bench_factor_solve(2))
    c.close()


def parse(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    acc = {}
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            key = next((k for k in KERNELS if k in name), None)
            if key == "k_big_schur" and "k_big_schur64" in name:
                key = "k_big_schur64"
            if key is None:
                continue
            a = acc.setdefault(key, {"dispatches": set()})
            a["dispatches"].add(row.get("Dispatch_Id", ""))
            a[row["Counter_Name"]] = a.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    out = {}
    for k, a in acc.items():
        n = len(a.pop("dispatches"))
        wc = a.get("SQ_WAVE_CYCLES", 0.0)
        r = {"dispatches": n, "per_dispatch": {c: v / n for c, v in a.items()}}
        if wc > 0:
            r["wait_any_over_wave_cycles"] = a.get("SQ_WAIT_ANY", 0.0) / wc
            r["wait_inst_any_over_wave_cycles"] = a.get("SQ_WAIT_INST_ANY", 0.0) / wc
            r["active_inst_any_over_wave_cycles"] = a.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
        if a.get("SQ_BUSY_CU_CYCLES", 0.0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in a:
            # cycles over SIMDs against quad-cycles over CUs: / (4 cycles per quad-cycle * 4 SIMDs per CU)
            r["mfma_busy_fraction_of_busy_cu_time"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (16.0 * a["SQ_BUSY_CU_CYCLES"])
        out[k] = r
    print(json.dumps({"source": os.path.basename(files[0]), "workload": "mat150: assembly x2, factorisation + solves x2 (no graph replay)", "kernels": out}, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "workload":
        workload(int(sys.argv[2]) if len(sys.argv) > 2 else 150)
    else:
        parse(sys.argv[2])
