"""Launch-by-launch timeline of ONE contact Newton iteration (everything between two k_contact_hessian launches) out of a rocprofv3 kernel-trace database,
followed by the launch counts of that iteration by kernel.
usage: python tools/contact_timeline.py <results.db> [which]"""
import collections
import sqlite3
import sys


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    n = n.replace("rocprim::ROCPRIM_400200_NS::detail::", "rocprim:").replace("ipcgpu::", "")
    return n[:60]


def main():
    db = sqlite3.connect(sys.argv[1])
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    rows = db.execute("select name,start,end,grid_x,workgroup_x from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if "k_contact_hessian" in r[0]]
    which = min(which, len(idx) - 2)
    seq = rows[idx[which]:idx[which + 1]]
    t0 = seq[0][1]
    prev_end = t0
    solver = ("k_big", "k_front", "k_extend", "k_xinv", "k_fwd", "k_bwd", "k_permute", "k_unpermute", "k_gather_a", "k_publish_flag", "k_entry", "k_scan_excl")
    cnt = collections.Counter()
    busy = 0
    for name, st, en, gx, wx in seq:
        n = short(name)
        cnt[n] += 1
        busy += en - st
        print(f"{n:60s} t={(st - t0) / 1e3:9.1f} gap={(st - prev_end) / 1e3:7.1f} dur={(en - st) / 1e3:7.1f} wgs={gx // max(wx, 1)}")
        prev_end = max(prev_end, en)
    total = (seq[-1][2] - t0) / 1e3
    ns = sum(c for k, c in cnt.items() if not k.startswith(solver))
    print(f"\niteration {which}: {len(seq)} launches ({ns} outside the solver), {total:.1f} us from first start to last end, {busy / 1e3:.1f} us of kernel time")
    for k, c in cnt.most_common():
        print(f"  {c:4d}  {k}")


if __name__ == "__main__":
    main()
