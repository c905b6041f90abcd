"""One TIME STEP (everything between two k_be_update launches) out of a rocprofv3 kernel-trace database: per kernel launches and time, the idle time between launches
(host round trips), and the launches in order with gaps above a threshold.
usage: python tools/step_timeline.py <results.db> [which] [gap_us]"""
import collections
import sqlite3
import sys


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    return n.replace("rocprim::ROCPRIM_400200_NS::detail::", "rocprim:").replace("ipcgpu::", "")[:48]


def main():
    db = sqlite3.connect(sys.argv[1])
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    thr = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
    rows = db.execute("select name,start,end from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if "k_be_update" in r[0] or "k_nm_update" in r[0]]
    which = min(which, len(idx) - 2)
    seq = rows[idx[which] + 1:idx[which + 1] + 1]
    t0 = seq[0][1]
    prev = t0
    agg = collections.OrderedDict()
    idle = 0.0
    for name, st, en in seq:
        n = short(name)
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += (en - st) / 1e3
        gap = (st - prev) / 1e3
        if gap > 0:
            idle += gap
        if gap > thr:
            print(f"  gap {gap:7.1f} us before {n} at t={(st - t0) / 1e3:9.1f}")
        prev = max(prev, en)
    total = (seq[-1][2] - t0) / 1e3
    print(f"time step {which}: {len(seq)} launches, {total:.1f} us, idle between launches {idle:.1f} us")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"  {c:4d} x {n:48s} {t:9.1f} us")


if __name__ == "__main__":
    main()
