"""Wave-cycle breakdown (SQ_WAIT_* / SQ_ACTIVE_INST_ANY / LDS counters as fractions of SQ_WAVE_CYCLES) per kernel out of a
rocprofv3 --pmc counter_collection.csv.  usage: python tools/pmc_wait_breakdown.py <dir>"""
import csv, glob, re, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(f)):
    mm = re.search(r"(k_[a-z_0-9]+)", row["Kernel_Name"])
    k = mm.group(1) if mm else row["Kernel_Name"][:24]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for kk in sorted(acc, key=lambda q: -acc[q].get("SQ_WAVE_CYCLES", 0))[:10]:
    c = acc[kk]
    wc = c.get("SQ_WAVE_CYCLES", 1) or 1
    print(f"{kk:22s} wave-cycles {wc:.3g}", {x.replace("SQ_", ""): round(v / wc, 3) for x, v in c.items() if x != "SQ_WAVE_CYCLES"}, "launches", n[kk])
