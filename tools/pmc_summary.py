"""Summarise a rocprofv3 --pmc counter_collection.csv for the render kernel (development aid)."""
import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
acc = {}
for r in rows:
    if "render_kernel" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-28s n=%d mean=%.6g" % (k, len(v), sum(v) / len(v)))
