"""Summarise a rocprofv3 --pmc counter_collection.csv (+kernel_trace.csv) per kernel."""
import csv, collections, glob, sys
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "rollout"
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if pat in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k)
        for c, vals in sorted(v.items()):
            print(f"   {c:28s} {sum(vals)/len(vals):16.1f}  (n={len(vals)})")
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    ts = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if pat in r["Kernel_Name"]]
    if ts:
        print(f"kernel durations us: n={len(ts)} mean={sum(ts)/len(ts):.1f} min={min(ts):.1f} max={max(ts):.1f}")
