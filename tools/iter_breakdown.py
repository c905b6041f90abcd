"""Where one Newton iteration of the bench spends its wall time, from a rocprofv3 kernel-trace csv (`--kernel-trace --output-format csv`).
An iteration = everything between two k_unpermute_x launches.  Kernels are put into classes; for each class: busy time (union of its intervals),
first start and last end relative to the iteration; plus the union over all kernels (GPU busy) against the iteration's wall time.
usage: python tools/iter_breakdown.py <dir with *kernel_trace.csv> [first] [count]"""
import csv
import glob
import os
import sys

CLASSES = [
    ("assembly", ("k_assemble_patch",)),
    ("factor: scatter / extend-add", ("k_gather_a", "k_scatter_big", "k_extend_add")),
    ("factor: fused fronts", ("k_front_fused",)),
    ("factor: 32-column steps", ("k_big_step",)),
    ("factor: Schur complements", ("k_big_schur",)),
    ("explicit inverses (side stream)", ("k_xinv_init", "k_xinv_gemm", "k_invert_blocks")),
    ("forward sweep", ("k_permute_rhs", "k_fwd_level", "k_big_fwd", "k_xinv_fwd")),
    ("backward sweep", ("k_bwd_level", "k_big_bwd", "k_xinv_bwd", "k_unpermute_x")),
]


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)


def main():
    f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
    rows.sort()
    idx = [i for i, r in enumerate(rows) if "k_unpermute_x" in r[2]]
    acc = {}
    walls, busys = [], []
    for it in range(first, first + count):
        seq = rows[idx[it] + 1:idx[it + 1] + 1]
        t0, t1 = rows[idx[it]][1], seq[-1][1]  # from the end of the previous solve to the end of this one
        walls.append(t1 - t0)
        busys.append(union([(s, e) for s, e, _, _ in seq]))
        other = []
        for cname, keys in CLASSES + [("everything else (energy, step filter, trial step, copies ...)", ())]:
            if keys:
                iv = [(s, e) for s, e, n, _ in seq if any(k in n for k in keys)]
            else:
                iv = [(s, e) for s, e, n, _ in seq if not any(k in n for _, ks in CLASSES for k in ks)]
            if not iv:
                continue
            a = acc.setdefault(cname, [0, 0, 0, 0])
            a[0] += union(iv)
            a[1] += min(s for s, _ in iv) - t0
            a[2] += max(e for _, e in iv) - t0
            a[3] += len(iv)
    n = count
    print(f"{os.path.basename(f)}: iterations {first}..{first + count - 1}; wall {sum(walls) / n / 1e3:.1f} us per iteration, GPU busy (union of all kernels) {sum(busys) / n / 1e3:.1f} us")
    print(f"{'class':66s} {'launches':>8s} {'busy us':>9s} {'first start':>12s} {'last end':>9s}")
    for cname, a in acc.items():
        print(f"{cname:66s} {a[3] / n:8.1f} {a[0] / n / 1e3:9.1f} {a[1] / n / 1e3:12.1f} {a[2] / n / 1e3:9.1f}")
    queues = sorted(set(q for _, _, _, q in rows))
    print("queues seen:", queues)


if __name__ == "__main__":
    main()
