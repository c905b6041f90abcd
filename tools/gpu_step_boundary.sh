cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/prof_b; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-contact --no-large --steps 60 > /dev/null 2>&1 )
db=$(find /tmp/prof_b -name "*.db" | head -1)
python - $db <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
def short(n): return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace("ipcgpu::", "")[:34]
idx = [i for i, r in enumerate(rows) if "k_be_update" in r[0]]
i = idx[5]
# from the last k_unpermute_x before the boundary to the first k_gather_a after it
j0 = max(k for k in range(i) if "k_unpermute_x" in rows[k][0])
j1 = min(k for k in range(i, len(rows)) if "k_gather_a" in rows[k][0])
t0 = rows[j0][1]; prev = rows[j0][2]
for name, st, en in rows[j0:j1 + 1]:
    print("%-34s t=%8.1f gap=%6.1f dur=%6.1f" % (short(name), (st - t0) / 1e3, (st - prev) / 1e3, (en - st) / 1e3)); prev = max(prev, en)
PY
