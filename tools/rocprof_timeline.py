"""Per-dispatch timeline of one factorisation + solve (everything between two k_unpermute_x launches) out of a rocprofv3
kernel-trace database.
usage: python tools/rocprof_timeline.py <results.db> [which]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    rows = db.execute("select name,start,end,grid_x,workgroup_x from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if "k_unpermute_x" in r[0]]
    seq = rows[idx[which] + 1:idx[which + 1] + 1]
    t0 = seq[0][1]
    prev_end = t0
    for name, st, en, gx, wx in seq:
        n = name.replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1].replace("void ", "")[:22]  # kernels live in an anonymous namespace
        print(f"{n:22s} t={(st - t0) / 1e3:8.1f} gap={(st - prev_end) / 1e3:5.1f} dur={(en - st) / 1e3:6.1f} wgs={gx // max(wx, 1)}")
        prev_end = en


if __name__ == "__main__":
    main()
