"""Summarise a rocprofv3 --kernel-trace --stats result database (rocpd sqlite) as a per-kernel table.
usage: python tools/rocprof_summary.py <results.db> [out.md]"""
import sqlite3
import sys


def short(name, n=110):
    name = name.replace("ipcgpu::(anonymous namespace)::", "").replace("void ", "")
    if name.startswith("Cijk_"):
        name = "rocBLAS/Tensile dgemm " + name[:40]
    return name.split("(")[0][:n]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    agg = {}
    for name, calls, tot, avg, pct in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += calls
        a[1] += tot
        a[2] += pct
    lines = ["| kernel | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
    for k, (calls, tot, pct) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        # the `top_kernels` view reports microseconds
        lines.append(f"| `{k}` | {calls} | {tot / 1e3:.3f} | {tot / calls:.2f} | {pct:.2f} |")
    out = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()
