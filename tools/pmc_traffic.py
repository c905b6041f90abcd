"""HBM traffic of the assembly kernel from the L2 memory-side counters (MI355X_MICROARCH.md, HBM / rocprofv3 PMC sections).

FETCH_SIZE and WRITE_SIZE do not fit one pass (TCC slots) and are uncalibrated on gfx950, so each pass also runs a copy
of known size (1 GiB device-to-device, the runtime's copy kernel) and the assembly kernel's counter is scaled by
known_bytes / counter(copy).  Run on the GPU box:

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_rd -- python $R/tools/pmc_traffic.py workload
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_wr -- python $R/tools/pmc_traffic.py workload
  python tools/pmc_traffic.py parse gpurun_out/pmc_rd gpurun_out/pmc_wr > profiles/r01_pmc_assembly_traffic.json
"""
import csv
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COPY_BYTES = 1 << 30


def workload(size=150):
    sys.path.insert(0, ROOT)
    import numpy as np
    from ipc_amd import lib, scene
    V, F = scene.make_mat(size)
    c = lib.Context(0)
    c.set_mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
    c.opt_init(0.04, False)
    left, right = scene.border_verts(V, 0.01)
    c.set_twist(left, right)
    c.set_positions(scene.twist_state(scene.jitter(V, F), 0.3))
    c.set_pattern()
    c.bench_stream(COPY_BYTES, 3)
    c.bench_assembly(0.04 ** 2, 5)
    c.close()


def counter_per_kernel(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    per = {}
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            per.setdefault((row["Kernel_Name"], row.get("Dispatch_Id", "")), 0.0)
            per[(row["Kernel_Name"], row.get("Dispatch_Id", ""))] += float(row["Counter_Value"])
    return per


def summarise(d, counter):
    per = counter_per_kernel(d, counter)
    copies = sorted(v for (k, _), v in per.items() if "copyBuffer" in k)
    asm = [v for (k, _), v in per.items() if "k_assemble_patch" in k and "true" in k]
    if not copies or not asm:
        raise SystemExit(f"kernels not found in {d}: {sorted(set(k for k, _ in per))[:8]}")
    big = copies[-3:]  # the three timed 1 GiB copies carry the largest values
    return sum(big) / len(big), sum(asm) / len(asm), len(asm)


def parse(rd_dir, wr_dir, size=150):
    sys.path.insert(0, ROOT)
    from ipc_amd import scene
    V, F = scene.make_mat(size)
    nT, nV = F.shape[0], V.shape[0]
    rd_copy, rd_asm, n1 = summarise(rd_dir, "FETCH_SIZE")
    wr_copy, wr_asm, n2 = summarise(wr_dir, "WRITE_SIZE")
    rd_scale, wr_scale = COPY_BYTES / rd_copy, COPY_BYTES / wr_copy
    out = {
        "kernel": "k_assemble_patch<true>", "workload": f"mat{size} ({nT} tets)", "launches_averaged": [n1, n2],
        "calibration": {"copy_bytes": COPY_BYTES, "FETCH_SIZE_per_copy": rd_copy, "WRITE_SIZE_per_copy": wr_copy,
                        "bytes_per_FETCH_SIZE_unit": rd_scale, "bytes_per_WRITE_SIZE_unit": wr_scale},
        "raw": {"FETCH_SIZE": rd_asm, "WRITE_SIZE": wr_asm},
        "read_bytes": rd_asm * rd_scale, "write_bytes": wr_asm * wr_scale,
        "traffic_bytes": rd_asm * rd_scale + wr_asm * wr_scale,
    }
    # B_asm = 112 nT + 84 nV + 8 nnz; nnz of the symmetric-upper CSR with 3x3 node blocks = 6 nV + 9 #edges
    edges = set()
    for a, b in ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)):
        lo, hi = np.minimum(F[:, a], F[:, b]), np.maximum(F[:, a], F[:, b])
        edges.update((lo.astype(np.int64) * nV + hi).tolist())
    out["algorithmic_bytes"] = 112 * nT + 84 * nV + 8 * (6 * nV + 9 * len(edges))
    out["traffic_over_algorithmic"] = out["traffic_bytes"] / out["algorithmic_bytes"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "workload":
        workload(int(sys.argv[2]) if len(sys.argv) > 2 else 150)
    else:
        parse(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 150)
