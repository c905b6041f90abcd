"""Launch-by-launch timeline of ONE Newton iteration of the bench out of a rocprofv3 kernel-trace csv (`--kernel-trace --output-format csv`):
kernel, queue, start relative to the iteration, duration, gap to the previous end on the same queue, workgroups.
An iteration = everything between two k_unpermute_x launches.
usage: python tools/iter_timeline.py <dir with *kernel_trace.csv> [which iteration]"""
import csv
import glob
import os
import sys


def main():
    f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            wg = max(int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1), 1)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // wg))
    rows.sort()
    idx = [i for i, r in enumerate(rows) if "k_unpermute_x" in r[2]]
    seq = rows[idx[which] + 1:idx[which + 1] + 1]
    t0 = rows[idx[which]][1]
    qs = sorted(set(r[3] for r in seq))
    lastEnd = {}
    print(f"iteration {which}: {len(seq)} launches, wall {(seq[-1][1] - t0) / 1e3:.1f} us; queues {qs}")
    for st, en, name, q, wgs in seq:
        n = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("::")[-1][:26]
        gap = (st - lastEnd[q]) / 1e3 if q in lastEnd else 0.0
        lastEnd[q] = en
        print(f"{n:26s} q{qs.index(q)} t={(st - t0) / 1e3:8.1f} dur={(en - st) / 1e3:6.1f} gap={gap:6.1f} wgs={wgs}")


if __name__ == "__main__":
    main()
